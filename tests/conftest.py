"""pytest configuration: markers, repo root on sys.path, fixture loaders shared by the suite."""
import glob
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NOISE_KIND = dict(WHITE_NOISE=1, GAUSSIAN_MIXTURE=2, GAUSSIAN_BIMODAL=3, ALPHA_STABLE=4, UNIFORM=5)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # The HIP library is a build artefact (git-ignored).  If this checkout has not been built yet, build it now: hipcc
    # cross-compiles gfx950 without a GPU.  On the GPU box the prebuilt .so travels with the snapshot and nothing happens.
    lib = os.path.join(ROOT, 'uncalibrated-visual-servoing_amd', 'libuvs_rmckf.so')
    if not os.path.exists(lib):
        import subprocess
        subprocess.run(['make', '-j', str(os.cpu_count() or 4), '-C', os.path.join(ROOT, 'uncalibrated-visual-servoing_amd', 'csrc')], check=True)


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + '*.npz')))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    d = {k: z[k] for k in z.files}
    d['meta'] = json.loads(str(d['meta']))
    return d


def scene_desired(g):
    """Features the fixture's SCENE projects onto at the goal pose (where the discs are): the servo target g['desired'] unless the fixture
    moved the target inside the same scene (oracle/gen_golden_estimators.py, *_target_shift)."""
    return g['scene_desired'] if 'scene_desired' in g else g['desired']


def rel_err(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


@pytest.fixture(scope='session')
def root():
    return ROOT


# Gate on the commanded dq for the rank-deficiency fixtures (default 1e-8).  Where a column is scaled by 1e12, numpy's SVD-based pinv
# (experiment.py:312) is itself only defined to ~cond * eps: moving every entry of the reference's own X by one ulp moves its command by
# 1.5e-3 (independent column) resp. 5e-4 (nearly parallel column) of its size -- measured in tests/test_oracle_golden.py.  The oracle and the
# kernels are held to that, not to 1e-8.
RANKDEF_CMD_TOL = {'rankdef_gmckf_scaled_1e12_indep': 1e-2, 'rankdef_gmckf_scaled_1e12_par_1em9': 3e-3}
