"""Oracle (test infrastructure): ctypes front of oracle/c/liboracle_rmckf.so, the plain-C restatement of the servo loop."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, 'c', 'liboracle_rmckf.so')
METHOD = {'KF': 2, 'MCKF': 3, 'IMCCKF': 4, 'GMCKF': 5}


class Params(C.Structure):
    _fields_ = [('m', C.c_int32), ('n', C.c_int32), ('method', C.c_int32), ('annealing', C.c_int32), ('k_max', C.c_int32),
                ('steps', C.c_int32), ('initial_guess', C.c_int32), ('pad', C.c_int32), ('kernel_bw', C.c_double),
                ('anneal_span', C.c_double), ('gain', C.c_double), ('dt', C.c_double), ('reg', C.c_double), ('desired', C.c_double * 32),
                ('fpi_threshold', C.c_double), ('fpi_epoch_max', C.c_int32), ('pad2', C.c_int32)]


class Plant(C.Structure):
    _fields_ = [('n_joints', C.c_int32), ('n_points', C.c_int32), ('theta_offset', C.c_double * 8), ('d', C.c_double * 8),
                ('a', C.c_double * 8), ('alpha', C.c_double * 8), ('points', (C.c_double * 3) * 16), ('focal', C.c_double), ('center', C.c_double),
                ('lin_J', C.c_void_p), ('lin_f0', C.c_void_p), ('lin_q0', C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(HERE, 'c', 'rmckf_oracle.c')
        if not os.path.exists(PATH) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(PATH)):
            subprocess.run(['make', '-C', os.path.join(HERE, 'c')], check=True)
        _lib = C.CDLL(PATH)
    return _lib


def ur10_plant():
    from . import plant_ref
    pl = Plant()
    pl.n_joints, pl.n_points = 6, 4
    for i, (off, d, a, alpha) in enumerate(plant_ref.DH_TABLE):
        pl.theta_offset[i], pl.d[i], pl.a[i], pl.alpha[i] = off, d, a, alpha
    for i, w in enumerate(plant_ref.place_discs()):
        for c in range(3):
            pl.points[i][c] = w[c]
    pl.focal, pl.center = plant_ref.FOCAL, plant_ref.CENTER
    return pl


def linear_plant(J, f0, q0):
    """f = f0 + J (q - q0): the consistent linearised camera of BASELINE config 5.  Keeps the arrays alive on the returned struct."""
    pl = Plant()
    J, f0, q0 = (np.ascontiguousarray(a, float) for a in (J, f0, q0))
    pl.n_joints, pl.n_points = J.shape[1], J.shape[0] // 2
    pl._keep = (J, f0, q0)
    pl.lin_J, pl.lin_f0, pl.lin_q0 = (a.ctypes.data for a in (J, f0, q0))
    return pl


def closed_loop_batch(q_start, noise, desired, method='GMCKF', kernel_bw=10.0, annealing=False, dt=0.05, t_max=15.0, gain=0.2,
                      steps=None, want_x=False, plant=None, fpi_threshold=0.1, fpi_epoch_max=1000, x0=None):
    """q_start (T, n), noise (T, K, m) -> dict(err (T,K,m), q (T,K,n), X (T,K,mn)?, stats (T,3), status, k_done, fpi (T,K) MCKF passes per step)."""
    q_start, noise = np.ascontiguousarray(q_start, float), np.ascontiguousarray(noise, float)
    T, K, m = noise.shape
    n = q_start.shape[1]
    fp = Params()
    fp.m, fp.n, fp.method, fp.annealing, fp.k_max = m, n, METHOD[method], int(annealing), int(t_max / dt)
    fp.steps, fp.initial_guess = K if steps is None else steps, int(x0 is None)
    fp.kernel_bw, fp.anneal_span, fp.gain, fp.dt, fp.reg = kernel_bw, 100.0, gain, dt, 0.001 ** 2
    fp.fpi_threshold, fp.fpi_epoch_max = fpi_threshold, fpi_epoch_max
    for i, v in enumerate(desired):
        fp.desired[i] = v
    pl = ur10_plant() if plant is None else plant
    err, q = np.zeros((T, K, m)), np.zeros((T, K, n))
    X = np.zeros((T, K, m * n)) if want_x else None
    stats, status, k_done = np.zeros((T, 3)), np.zeros(T, np.int32), np.zeros(T, np.int32)
    fpi = np.zeros((T, fp.steps), np.int32)
    ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)     # noqa: E731
    if x0 is not None:                                            # supplied X0 (the same for every trial or one row per trial): one trial at a time
        x0 = np.ascontiguousarray(np.broadcast_to(np.asarray(x0, float).reshape(-1, m * n), (T, m * n)))
        one = lib().uvs_oracle_closed_loop
        one.restype, one.argtypes = C.c_int, [C.POINTER(Params), C.POINTER(Plant)] + [C.c_void_p] * 9
        at = lambda a, t: None if a is None else C.c_void_p(a[t:t + 1].ctypes.data)     # noqa: E731
        for t in range(T):
            status[t] = one(C.byref(fp), C.byref(pl), at(q_start, t), at(noise, t), at(x0, t), at(err, t), at(q, t), at(X, t), at(stats, t), at(k_done, t), at(fpi, t))
        return dict(err=err, q=q, X=X, stats=stats, status=status, k_done=k_done, fpi=fpi)
    fn = lib().uvs_oracle_closed_loop_batch
    fn.restype, fn.argtypes = None, [C.POINTER(Params), C.POINTER(Plant), C.c_int64] + [C.c_void_p] * 9
    fn(C.byref(fp), C.byref(pl), T, ptr(q_start), ptr(noise), ptr(err), ptr(q), ptr(X), ptr(stats), ptr(status), ptr(k_done), ptr(fpi))
    return dict(err=err, q=q, X=X, stats=stats, status=status, k_done=k_done, fpi=fpi)


def replay_batch(f_seq, dq_seq, x0, desired, method='GMCKF', kernel_bw=10.0, annealing=False, k_max=300, gain=0.2, fpi_threshold=0.1, fpi_epoch_max=1000):
    """f_seq (T, K+1, m), dq_seq (T, K, n), x0 (T, m n) -> dict(X (T,K,mn), dq_cmd (T,K,n), kappa (T,K,m), status, k_done, fpi (T,K))."""
    f_seq, dq_seq, x0 = (np.ascontiguousarray(a, float) for a in (f_seq, dq_seq, x0))
    T, K, n = dq_seq.shape
    m = f_seq.shape[2]
    fp = Params()
    fp.m, fp.n, fp.method, fp.annealing, fp.k_max, fp.steps, fp.initial_guess = m, n, METHOD[method], int(annealing), int(k_max), K, 0
    fp.kernel_bw, fp.anneal_span, fp.gain, fp.dt, fp.reg = kernel_bw, 100.0, gain, 0.0, 0.001 ** 2
    fp.fpi_threshold, fp.fpi_epoch_max = fpi_threshold, fpi_epoch_max
    for i, v in enumerate(desired):
        fp.desired[i] = v
    X, cmd, kap = np.zeros((T, K, m * n)), np.zeros((T, K, n)), np.zeros((T, K, m))
    status, k_done, fpi = np.zeros(T, np.int32), np.zeros(T, np.int32), np.zeros((T, K), np.int32)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)      # noqa: E731
    fn = lib().uvs_oracle_replay_batch
    fn.restype, fn.argtypes = None, [C.POINTER(Params), C.c_int64] + [C.c_void_p] * 9
    fn(C.byref(fp), T, ptr(f_seq), ptr(dq_seq), ptr(x0), ptr(X), ptr(cmd), ptr(kap), ptr(status), ptr(k_done), ptr(fpi))
    return dict(X=X, dq_cmd=cmd, kappa=kap, status=status, k_done=k_done, fpi=fpi)
