"""bench.py run as the driver runs it (a subprocess, stdout and stderr captured), at reduced table sizes: the LAST stdout line is
the bounded contract line, every leg is in bench_detail.json, and stderr carries no Python warning (VERDICT r4 missing #1)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_subprocess_last_stdout_line_is_the_contract_line(tmp_path):
    env = dict(os.environ, PYTHONWARNINGS='default')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--users', '2000001',
           '--items-per-domain', '1000000', '--batch', '65536', '--cpu-seconds', '0.5', '--no-config-legs']
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    last = lines[-1]
    assert len(last) <= 4096, len(last)
    d = json.loads(last)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'cpu_baseline', 'bench_wall_s', 'detail_file'):
        assert k in d, k
    assert d['steps'] == 3 and d['warmup'] == 1 and d['n_gpus'] == 1 and d['vs_baseline'] is None
    assert abs(d['value'] - 2 * 65536 / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-6
    r = d['roofline']
    assert r['bound'] == 'hbm' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and r['peak'] == 8000.0
    assert r['traffic'] is None or isinstance(r['traffic'], int)
    assert d['cpu_baseline']['kind'] == 'port' and d['cpu_baseline']['value'] > 0
    full = json.load(open(os.path.join(ROOT, d['detail_file'])))
    assert full['value'] == d['value'] and 'kernels' in full and 'e2e' in full
    bad = [l for l in p.stderr.splitlines() if 'Warning' in l or 'Traceback' in l]
    assert not bad, bad[:5]
