"""Import shim: the product package lives in the hyphenated directory ``recbole-cdr_amd/`` (not a valid Python
identifier), so ``import recbole_cdr_amd`` loads this file, which registers that directory as the package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'recbole-cdr_amd')
_spec = importlib.util.spec_from_file_location('recbole_cdr_amd', os.path.join(_dir, '__init__.py'),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['recbole_cdr_amd'] = _mod
_spec.loader.exec_module(_mod)
