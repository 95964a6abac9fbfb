"""bench.py keeps the driver's contract: one JSON line with the agreed keys (GPU), and a CPU baseline that sizes itself honestly (CPU)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_available_cores_respects_affinity_and_quota():
    import bench
    n = bench.available_cores()
    assert 1 <= n <= len(os.sched_getaffinity(0))


def test_cpu_baseline_runs_the_oracle_on_a_small_sample():
    import bench
    q = np.tile(np.array([0.0, 0.0, 1.96349541, 0.0, -1.57079633, 0.0]), (2, 1))
    out = bench.cpu_baseline(q, budget_trials_per_core=1)
    assert out['kind'] == 'port' and out['unit'] == 'updates/s' and out['cores'] == bench.available_cores()
    assert out['value'] > 0 and 'oracle/rmckf_dense.py' in out['sample']


def test_rocm_smi_text_is_parsed():
    import bench
    text = """
============================ ROCm System Management Interface ============================
GPU[0]		: fclk clock level: 0: (1250Mhz)
GPU[0]		: mclk clock level: 0: (2000Mhz)
GPU[0]		: sclk clock level: 1: (2131Mhz)
======================================= Power Cap ========================================
GPU[0]		: Max Graphics Package Power (W): 1400.0
=================================== Power Consumption ====================================
GPU[0]		: Current Socket Graphics Package Power (W): 1357.0
"""
    assert bench.parse_rocm_smi(text) == (1357.0, 2131, 1400.0)
    assert bench.parse_rocm_smi('GPU[0] : sclk clock level: S: (95Mhz)\nGPU[0] : Average Graphics Package Power (W): 246.0') == (246.0, 95, None)
    assert bench.parse_rocm_smi('no such tool') == (None, None, None)


@pytest.mark.gpu
def test_bench_line_contract():
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', '--trials', '4096', '--no-cpu-baseline']
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1                                                   # exactly one JSON line on stdout
    d = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
                'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in d, key
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['higher_is_better'] is True and d['scaling'] == 'strong'
    assert d['config']['trials_total'] == 4096 and d['config']['ranks_seen'] == 1 and d['multi_gpu']['gather_inside_timed_region'] is False
    assert d['vs_baseline'] is None and d['dtype'] == 'f64' and d['data'] == 'synthetic' and d['unit'] == 'updates/s'
    assert 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    assert d['value'] > 0 and abs(d['value'] - 4096 * 299 / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-6
    assert 'power' in r and (r['power'] is None or (len(r['power']['sclk_mhz']) >= 1 and min(r['power']['package_watts']) > 0))
    assert d['replay']['estimator_only']['achieved'] > 0
