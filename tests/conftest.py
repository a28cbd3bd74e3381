import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_addoption(parser):
    parser.addoption('--slow-gpu', action='store_true', default=False,
                     help='also run the tests marked slow_gpu (bench.py subprocess runs of 20-45 s each; `-m slow_gpu` selects them alone)')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run on the GPU box via gpurun)')
    config.addinivalue_line('markers', 'slow_gpu: a GPU test of more than ~20 s that is NOT a parity test (bench.py subprocess / fallback-chain runs); '
                                       'left out of the default `-m gpu` selection, run with --slow-gpu or -m slow_gpu')


def pytest_collection_modifyitems(config, items):
    # the default `-m gpu` selection stays well inside the driver's time limit: slow_gpu tests only on request (every parity test is in the default)
    if not config.getoption('--slow-gpu') and 'slow_gpu' not in (config.getoption('-m') or ''):
        keep, drop = [], []
        for item in items:
            (drop if 'slow_gpu' in item.keywords else keep).append(item)
        if drop:
            config.hook.pytest_deselected(items=drop)
            items[:] = keep
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
