"""Regression pin of the library against itself: bit-level digests of closed-loop outputs taken from the round-4 kernels
(tools/golden_digest.py --write, tests/golden/digests.json).  A kernel rewrite that claims "same arithmetic, same bits" -- round 5: the MCKF
fixed-point branch spread over the wavefront -- has to reproduce them: X, err, q streams (logged rows), statistics, status, k_done, at
4 099 / 40 000 (segmented) / 65 536 trials, Cauchy noise included.  This is NOT parity with the reference (tests/test_gpu_parity.py,
tests/test_gpu_mckf_fpi.py hold that); it is the guarantee that a rewrite did not move a bit."""
import json
import os
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import golden_digest  # noqa: E402

WANT = json.load(open(golden_digest.PATH))['cases']


@pytest.mark.parametrize('case', golden_digest.CASES, ids=[c[0] for c in golden_digest.CASES])
def test_outputs_keep_their_bits(case):
    import torch
    import bench
    import uvs_amd
    name, got = golden_digest.run_case(uvs_amd, torch, bench, case)
    torch.cuda.empty_cache()
    assert name in WANT, 'no digest on file: run tools/golden_digest.py --write on a GPU box BEFORE changing the kernels'
    assert got['noise'] == WANT[name]['noise'], 'the INPUT changed (noise generator), not the estimator'
    diff = {k: (got[k], WANT[name][k]) for k in got if got[k] != WANT[name][k]}
    assert not diff, diff
