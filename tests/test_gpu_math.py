"""Device fp64 helper functions (csrc/rmckf_math.hpp) against numpy, through the uvs_debug_math_f64 test hook."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _eval(which, x):
    import torch
    import uvs_amd
    xd = torch.as_tensor(np.ascontiguousarray(x, float), device='cuda')
    yd = torch.empty_like(xd)
    rc = uvs_amd.lib().uvs_debug_math_f64(which, xd.numel(), xd.data_ptr(), yd.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    uvs_amd._lib.check(rc)
    return yd.cpu().numpy()


def _ulps(got, ref):
    return np.abs(got - ref) / np.spacing(np.abs(ref))


def test_fast_reciprocal_and_roots():
    rng = np.random.default_rng(0)
    x = np.exp(rng.uniform(np.log(1e-150), np.log(1e150), 200000)) * rng.choice([-1.0, 1.0], 200000)
    assert _ulps(_eval(0, x), 1.0 / x).max() <= 1.5
    xp = np.abs(x)
    assert _ulps(_eval(1, xp), np.sqrt(xp)).max() <= 1.0
    assert _ulps(_eval(2, xp), 1.0 / np.sqrt(xp)).max() <= 2.5


def test_sincos_bounded_and_fallback():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-12, 12, 200000), rng.uniform(-1e5, 1e5, 100000), np.arange(-40, 41) * (np.pi / 2),
                        np.arange(-40, 41) * (np.pi / 2) + 1e-9, [0.0, 1e-300, -1e-300, 1e-8], rng.uniform(-1e12, 1e12, 2000)])
    s, c = _eval(3, x), _eval(4, x)
    # absolute error relative to 1 (what the kinematics sees) and relative error away from the zeros
    assert np.abs(s - np.sin(x)).max() <= 2.3e-16 and np.abs(c - np.cos(x)).max() <= 2.3e-16
    big = np.abs(np.sin(x)) > 1e-3
    assert _ulps(s[big], np.sin(x)[big]).max() <= 2.0
    big = np.abs(np.cos(x)) > 1e-3
    assert _ulps(c[big], np.cos(x)[big]).max() <= 2.0


def test_exp_on_kernel_range():
    x = -np.exp(np.random.default_rng(2).uniform(np.log(1e-12), np.log(700), 200000))
    assert _ulps(_eval(5, x), np.exp(x)).max() <= 2.0
    assert _ulps(_eval(6, x), np.exp(x)).max() <= 2.0                       # exp_nonpos, the form the tuned kernel uses
    edge = np.array([0.0, -0.0, -1e-300, -745.0, -746.0, -1000.0, -1e300, -np.inf])
    got = _eval(6, edge)
    assert got[0] == 1.0 and got[1] == 1.0 and got[2] == 1.0 and np.all(got[4:] == 0.0) and abs(got[3] / np.exp(-745.0) - 1) < 1e-3
    assert np.isnan(_eval(6, np.array([np.nan]))[0])


def test_log_own_routine_and_fallback():
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(0, 1, 300000), np.exp(rng.uniform(-708, 709, 300000)), -np.log(rng.uniform(0, 1, 200000)),
                        1.0 + rng.uniform(-0.5, 0.5, 200000) * 10.0 ** rng.uniform(-15, 0, 200000),
                        np.cos(rng.uniform(-np.pi / 2, np.pi / 2, 200000)), [1.0, 2.0, 0.5, 2.2250738585072014e-308, 1.7976931348623157e308]])
    x = x[(x >= 2.2250738585072014e-308) & np.isfinite(x)]
    got, ref = _eval(7, x), np.log(x)
    nz = ref != 0
    assert _ulps(got[nz], ref[nz]).max() <= 1.5 and np.all(got[~nz] == 0.0)
    # zero, subnormal, negative, non-finite: the library's results (numpy's semantics)
    edge = np.array([0.0, -0.0, 5e-324, 1e-310, -1.0, np.inf, -np.inf, np.nan])
    with np.errstate(all='ignore'):
        want = np.log(edge)
    got = _eval(7, edge)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    fin = ~np.isnan(want)
    assert np.all((got[fin] == want[fin]) | (_ulps(got[fin], want[fin]) <= 1.0))


def test_exp_clamped_full_range():
    rng = np.random.default_rng(4)
    x = np.concatenate([rng.uniform(-745, 709.7, 300000), rng.uniform(-1, 1, 100000) * 10.0 ** rng.uniform(-12, 0, 100000)])
    assert _ulps(_eval(8, x), np.exp(x)).max() <= 2.0
    edge = np.array([0.0, 709.78, 709.79, 800.0, 1e300, np.inf, -746.0, -1e300, -np.inf])
    got = _eval(8, edge)
    with np.errstate(all='ignore'):
        want = np.exp(edge)
    assert got[0] == 1.0 and abs(got[1] / want[1] - 1) < 1e-14 and np.all(np.isinf(got[2:6])) and np.all(got[6:] == 0.0)
    assert np.isnan(_eval(8, np.array([np.nan]))[0])


def test_incremental_sincos_both_tiers():
    """sin / cos carried by the addition theorems (rmckf_math.hpp: sincos_advance for joint steps up to 0.1 rad, sincos_advance_wide up to
    1 rad -- config 3's held outliers): a few ulp of the pair's own magnitude, measured against numpy at the angle 0.7."""
    rng = np.random.default_rng(3)
    for which_s, which_c, bound in ((9, 10, 0.1), (11, 12, 1.0)):
        d = np.concatenate([rng.uniform(-bound, bound, 200000), [0.0, bound, -bound, 1e-300, -1e-9]])
        s, c = _eval(which_s, d), _eval(which_c, d)
        # the pair starts 1 ulp off the true sin / cos of 0.7 (decimal constants); what is bounded is the absolute error on the unit circle
        assert np.abs(s - np.sin(0.7 + d)).max() <= 4e-16 and np.abs(c - np.cos(0.7 + d)).max() <= 4e-16
        assert np.abs(s * s + c * c - 1.0).max() <= 6e-16
