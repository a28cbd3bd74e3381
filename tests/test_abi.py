"""CPU-side checks of the boundary: the library builds/loads here and exports every symbol include/cdr_hip.h declares;
the product refuses CPU tensors instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

import recbole_cdr_amd
from recbole_cdr_amd import binding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'cdr_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(cdr_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    if not os.path.isfile(binding.lib_path()):
        binding.build()
    lib = ctypes.CDLL(binding.lib_path())
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in include/cdr_hip.h but not exported'
    assert sorted(binding.exported_symbols()) == syms, 'binding.py signatures out of sync with the header'
    assert binding.load().cdr_abi_version() == binding.ABI_VERSION


def test_no_cpu_fallback():
    from recbole_cdr_amd import functional as F_
    w = torch.randn(8, 4)
    ids = torch.tensor([1, 2])
    with pytest.raises(binding.NativeLibraryError):
        F_.gather_rows(w, ids)
    with pytest.raises(binding.NativeLibraryError):
        F_.BPRGatherLoss.apply(w, w, ids, ids, ids, 1e-10, 0.0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'recbole-cdr_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f'{f} imports the oracle'


def test_get_model_lookup():
    assert recbole_cdr_amd.get_model('EMCDR').__name__ == 'EMCDR'
    assert recbole_cdr_amd.get_model('CMF').__name__ == 'CMF'
    with pytest.raises(ValueError):
        recbole_cdr_amd.get_model('NoSuchModel')


def test_header_is_valid_c_and_c_consumer_links():
    """include/cdr_hip.h compiles as C11 (gcc -fsyntax-only) and the plain-C consumer tests/abi_c/abi_smoke.c builds against the
    library -- the boundary does not depend on C++ or torch.  (The program itself runs in the -m gpu suite.)"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    probe = '#include "cdr_hip.h"\nint main(void) { return cdr_abi_version() == CDR_ABI_VERSION ? 0 : 1; }\n'
    r = subprocess.run(['gcc', '-std=c11', '-Wall', '-Werror', '-fsyntax-only', '-x', 'c', '-', '-I', os.path.join(root, 'include')],
                       input=probe.encode(), capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    r = subprocess.run(['make', '-B', '-C', os.path.join(root, 'tests', 'abi_c')], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    assert os.path.isfile(os.path.join(root, 'tests', 'abi_c', 'abi_smoke'))
