"""Typed Python front of the C ABI: builds ``uvs_filter_params``, owns nothing but torch tensors on the GPU,
and launches the HIP kernels on torch's current stream.  No numpy arithmetic of the estimator lives here --
if the library or the GPU is missing these calls raise.

Buffer layout.  Per-step streams are allocated *trial-fastest*: ``[step][component][trial]`` (``layout='kct'``).
With one filter per lane (lanes_per_filter = 1) consecutive lanes then touch consecutive doubles, so every
wavefront load/store of a stream component is one contiguous 512-byte segment.  ``layout='ktc'``
(``[step][trial][component]``) suits the variants where several lanes share a filter; ``'tkc'`` is the
reference's per-trial record order.  The kernels take strides, so all three work everywhere.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import FilterParams, View, NULL_VIEW

METHOD_CODES = {'KF': 2, 'MCKF': 3, 'IMCCKF': 4, 'GMCKF': 5}
REG = 0.001 ** 2            # experiment.py:280
ANNEAL_SPAN = 100.0         # experiment.py:271


def loop_clock(t_s, t_max):
    """Times at which the reference loop body runs: the clock starts at t_s (ur10_simulation.py:57) and advances by
    t_s per iteration while t < t_max (experiment.py:125), accumulated in floating point exactly like the simulator stub."""
    ts, t = [], 0.0 + t_s
    while t < t_max:
        ts.append(t)
        t += t_s
    return np.array(ts)


def make_params(m, n, method='GMCKF', kernel_bw=10.0, annealing=False, t_s=0.05, t_max=15.0, gain=0.2, desired=None,
                initial_guess=True, lanes=0, steps=None, fpi_threshold=0.1, fpi_epoch_max=1000):
    code = METHOD_CODES[method] if isinstance(method, str) else int(getattr(method, 'value', method))
    fp = FilterParams()
    fp.m, fp.n, fp.method, fp.annealing = m, n, code, int(bool(annealing))
    fp.k_max = int(t_max / t_s)                                     # experiment.py:120
    fp.steps = len(loop_clock(t_s, t_max)) if steps is None else int(steps)
    fp.initial_guess, fp.lanes_per_filter = int(bool(initial_guess)), int(lanes)
    fp.kernel_bw, fp.anneal_span, fp.gain, fp.dt, fp.reg = float(kernel_bw), ANNEAL_SPAN, float(gain), float(t_s), REG
    fp.fpi_threshold, fp.fpi_epoch_max = float(fpi_threshold), int(fpi_epoch_max)
    if m > _lib.UVS_MAX_M or n > _lib.UVS_MAX_N:
        raise ValueError('(m, n) exceeds UVS_MAX_M / UVS_MAX_N')
    if desired is not None:
        for i, v in enumerate(np.asarray(desired, float).ravel()):
            fp.desired[i] = v
    return fp


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise _lib.UvsLibraryError('no GPU visible: the RMCKF path runs only on the HIP library (no CPU fallback)')
    return torch


def _stream():
    return C.c_void_p(_torch().cuda.current_stream().cuda_stream)


_DIMS = {'kct': (2, 0, 1), 'ktc': (1, 0, 2), 'tkc': (0, 1, 2)}


def alloc_stream(T, K, comp, layout='kct', device='cuda', zero=False):
    """fp64 tensor for a [trial][step][comp] stream in the requested physical layout.

    Experiment knob UVS_ROW_PAD=<trials>: pitch trial-fastest rows at T + pad (the [:, :, :T] view is returned; every entry point takes
    strides).  With dense rows of a power-of-two T every (step, component) row starts in the same 256-byte slot of the address interleave
    and two of the eight slots drain 17 % slower on MI355X; walking the slots from row to row measured -1 % (config 3), -2 % (IMCC-KF),
    +/-0 (headline) and +8 % (replay) on one box -- DESIGN.md appendix A.1 -- so dense rows stay the default."""
    torch = _torch()
    pad = int(os.environ.get('UVS_ROW_PAD', '0'))
    if layout == 'kct' and pad:
        return (torch.zeros if zero else torch.empty)((K, comp, T + pad), dtype=torch.float64, device=device)[:, :, :T]
    shape = {'kct': (K, comp, T), 'ktc': (K, T, comp), 'tkc': (T, K, comp)}[layout]
    return (torch.zeros if zero else torch.empty)(shape, dtype=torch.float64, device=device)


def stream_view(tensor, layout='kct'):
    return NULL_VIEW if tensor is None else _lib.view_of(tensor, _DIMS[layout])


def as_tkc(tensor, layout='kct'):
    """Logical [trial][step][comp] view (no copy) of a stream tensor."""
    return tensor.permute({'kct': (2, 0, 1), 'ktc': (1, 0, 2), 'tkc': (0, 1, 2)}[layout])


def supported_lanes(m, n):
    buf = (C.c_int32 * 16)()
    cnt = _lib.lib().uvs_supported_lanes(m, n, buf, 16)
    return [buf[i] for i in range(min(cnt, 16))]


_WORKSPACES = {}


def workspace(fp, plant_struct, T, device):
    """(pointer, bytes) of the scratch buffer uvs_rmckf_closed_loop_ws_f64 wants for this launch -- (None, 0) when it wants none.  One
    buffer per device and stream, grown on demand and kept: the library allocates nothing itself."""
    need = int(_lib.lib().uvs_rmckf_closed_loop_workspace_bytes(C.byref(fp), C.byref(plant_struct), T))
    if need == 0:
        return None, 0
    torch = _torch()
    key = (str(device), torch.cuda.current_stream().cuda_stream)
    buf = _WORKSPACES.get(key)
    if buf is None or buf.numel() < need:
        buf = _WORKSPACES[key] = torch.empty(need, dtype=torch.uint8, device=device)
    return buf.data_ptr(), need


def hand_over_fallbacks(fp, plant_struct, T, device=None):
    """Work items of the LAST segmented launch of this (fp, plant, T) on this stream that ran out their spin budget and recomputed their trial
    from step 0 (uvs_rmckf_closed_loop_fallback_offset): 0 on a healthy launch, None when the launch is not segmented.  Synchronises."""
    torch = _torch()
    off = int(_lib.lib().uvs_rmckf_closed_loop_fallback_offset(C.byref(fp), C.byref(plant_struct), T))
    if off == 0:
        return None
    device = device if device is not None else torch.device('cuda', torch.cuda.current_device())
    buf = _WORKSPACES.get((str(device), torch.cuda.current_stream().cuda_stream))
    if buf is None:
        return None
    return int(buf[off:off + 4].view(torch.int32).item())


def launch_closed_loop(fp, plant_struct, T, *args, device=None):
    """uvs_rmckf_closed_loop_ws_f64 on torch's current stream with this process's cached workspace: ``args`` are the views / pointers of
    uvs_rmckf_closed_loop_f64 between ``T`` and ``stream``, in the header's order.  Returns the library's return code."""
    torch = _torch()
    ws, ws_bytes = workspace(fp, plant_struct, T, device if device is not None else torch.device('cuda', torch.cuda.current_device()))
    return _lib.lib().uvs_rmckf_closed_loop_ws_f64(C.byref(fp), C.byref(plant_struct), T, *args, ws, ws_bytes, _stream())


def closed_loop(fp, plant_struct, q_start, noise=None, x0=None, want=('x', 'err', 'q'), layout='kct', final_state=False, x_layout=None, reuse=None):
    """Launch T closed-loop trials.  ``q_start``: (T, n) cuda tensor; ``noise``: stream tensor in ``layout`` or None;
    ``x0``: (T, m*n) cuda tensor when fp.initial_guess == 0.  Returns a dict of output tensors (streams in ``layout``).
    ``x_layout``: another layout for the X stream alone.  The default -- X trial-fastest like every stream -- is what the kernels are tuned for (since round 6
    the (8,6) KF / IMCC-KF kernels write it as 16-byte pairs of consecutive trials when T is even).  'ktc' (per-trial records) is faster still for KF on
    batches above 16 384 trials (two lanes per filter; bench.py `other_estimators.KF.x_records`: up to - 10 %, box-dependent) and level with the default for
    IMCC-KF.  Only there: RMCKF and MCKF keep their strided stores whatever the view, and smaller batches run on the four-lane kernels, for which 'ktc' is an
    uncoalesced, slower path.
    ``reuse``: the dict an earlier call with at least as many trials returned -- its tensors are written again ([..., :T] of the streams,
    [:T] of the per-trial arrays) instead of allocating new ones (batch.run_sweep: cell after cell through one set of buffers)."""
    x_layout = x_layout or layout
    torch = _torch()
    T, K, m, n = q_start.shape[0], fp.steps, fp.m, fp.n
    dev = q_start.device
    out = {}
    tdim = lambda lay: {'kct': 2, 'ktc': 1, 'tkc': 0}[lay]         # noqa: E731
    for key, comp in (('x', m * n), ('err', m), ('q', n), ('f', m), ('dq', n)):
        lay = x_layout if key == 'x' else layout
        if key not in want:
            out[key] = None
        elif reuse is not None:
            out[key] = reuse[key].narrow(tdim(lay), 0, T)
        else:
            out[key] = alloc_stream(T, K, comp, lay, dev)          # rows at and after k_done are unspecified
    if reuse is not None:
        out['stats'], out['status'], out['k_done'] = reuse['stats'][:T], reuse['status'][:T], reuse['k_done'][:T]
    else:
        out['stats'] = torch.zeros((T, 3), dtype=torch.float64, device=dev)
        out['status'] = torch.zeros(T, dtype=torch.int32, device=dev)
        out['k_done'] = torch.zeros(T, dtype=torch.int32, device=dev)
    out['x_final'] = torch.empty((T, m * n), dtype=torch.float64, device=dev) if final_state else None
    out['p_final'] = torch.empty((T, m * n * n), dtype=torch.float64, device=dev) if final_state else None
    flat = lambda t: NULL_VIEW if t is None else View(t.data_ptr(), t.stride(0), 0, t.stride(1))      # noqa: E731
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)           # around the launch only: the
    ws, ws_bytes = workspace(fp, plant_struct, T, dev)
    start.record()                                                                                     # allocations above are not kernel time
    rc = _lib.lib().uvs_rmckf_closed_loop_ws_f64(
        C.byref(fp), C.byref(plant_struct), T, flat(q_start), stream_view(noise, layout), flat(x0),
        stream_view(out['x'], x_layout), stream_view(out['err'], layout), stream_view(out['q'], layout),
        stream_view(out['f'], layout), stream_view(out['dq'], layout),
        out['stats'].data_ptr(), out['status'].data_ptr(), out['k_done'].data_ptr(),
        flat(out['x_final']), flat(out['p_final']), ws, ws_bytes, _stream())
    stop.record()
    _lib.check(rc)
    out['events'] = (start, stop)
    return out


def replay(fp, f, dq, x0, want=('x', 'err', 'kappa', 'dqcmd'), layout='kct', final_state=False, in_layout=None):
    """Open-loop replay.  ``f``: stream tensor with K+1 steps, ``dq``: K steps, ``x0``: (T, m*n).  ``layout``: physical layout of the
    output streams and, unless ``in_layout`` says otherwise, of ``f`` / ``dq``."""
    in_layout = in_layout or layout
    torch = _torch()
    T, K, m, n = x0.shape[0], fp.steps, fp.m, fp.n
    dev = x0.device
    out = {}
    for key, comp in (('x', m * n), ('err', m), ('kappa', m), ('dqcmd', n)):
        out[key] = alloc_stream(T, K, comp, layout, dev) if key in want else None      # rows at and after k_done are unspecified
    out['status'] = torch.zeros(T, dtype=torch.int32, device=dev)
    out['k_done'] = torch.zeros(T, dtype=torch.int32, device=dev)
    out['x_final'] = torch.empty((T, m * n), dtype=torch.float64, device=dev) if final_state else None
    out['p_final'] = torch.empty((T, m * n * n), dtype=torch.float64, device=dev) if final_state else None
    flat = lambda t: NULL_VIEW if t is None else View(t.data_ptr(), t.stride(0), 0, t.stride(1))      # noqa: E731
    rc = _lib.lib().uvs_rmckf_replay_f64(
        C.byref(fp), T, stream_view(f, in_layout), stream_view(dq, in_layout), flat(x0),
        stream_view(out['x'], layout), stream_view(out['err'], layout), stream_view(out['kappa'], layout),
        stream_view(out['dqcmd'], layout), out['status'].data_ptr(), out['k_done'].data_ptr(),
        flat(out['x_final']), flat(out['p_final']), _stream())
    _lib.check(rc)
    return out


def replay_f32(fp, f, dq, x0, want=('x', 'err'), layout='kct'):
    """Single-precision estimator-only replay (uvs_rmckf_replay_f32: a measured lower-precision variant, never the parity path).
    ``f`` (K + 1 steps), ``dq`` (K steps): fp32 stream tensors in ``layout``; ``x0``: (T, m*n) fp32."""
    torch = _torch()
    assert f.dtype == dq.dtype == x0.dtype == torch.float32
    T, K, m, n = x0.shape[0], fp.steps, fp.m, fp.n
    dev = x0.device
    shape = lambda c: {'kct': (K, c, T), 'ktc': (K, T, c), 'tkc': (T, K, c)}[layout]      # noqa: E731
    out = {'x': torch.empty(shape(m * n), dtype=torch.float32, device=dev) if 'x' in want else None,
           'err': torch.empty(shape(m), dtype=torch.float32, device=dev) if 'err' in want else None,
           'status': torch.zeros(T, dtype=torch.int32, device=dev), 'k_done': torch.zeros(T, dtype=torch.int32, device=dev)}
    rc = _lib.lib().uvs_rmckf_replay_f32(C.byref(fp), T, stream_view(f, layout), stream_view(dq, layout), View(x0.data_ptr(), x0.stride(0), 0, x0.stride(1)),
                                         stream_view(out['x'], layout), stream_view(out['err'], layout), out['status'].data_ptr(), out['k_done'].data_ptr(), _stream())
    _lib.check(rc)
    return out


class FilterBank:
    """T estimators whose state (X, P) stays in HBM between ``step`` calls: the drop-in used when the robot is external."""

    def __init__(self, fp, T=1, x0=None, device='cuda'):
        torch = _torch()
        self.fp, self.T = fp, T
        m, n = fp.m, fp.n
        self.X = torch.zeros((T, m * n), dtype=torch.float64, device=device)
        if x0 is not None:
            self.X.copy_(torch.as_tensor(np.asarray(x0, float).reshape(T, m * n)))
        self.P = torch.eye(n, dtype=torch.float64, device=device).repeat(T, m, 1, 1).contiguous()   # P = I (experiment.py:73)
        self.dq = torch.zeros((T, n), dtype=torch.float64, device=device)
        self.err = torch.zeros((T, m), dtype=torch.float64, device=device)
        self.kappa = torch.ones((T, m), dtype=torch.float64, device=device)
        self.status = torch.zeros(T, dtype=torch.int32, device=device)
        self.first = True
        self._host = None

    def _host_io(self):
        """Pinned host records the step kernel reads and writes in place (zero-copy: hipHostMalloc memory is mapped into the device's address
        space at the same address): two input records [f | f_old] and two output records [dq | err | kappa], status words, used alternately so
        that call k reads the command of call k - 1 as its regressor.  numpy views are made once."""
        torch = _torch()
        m, n, T = self.fp.m, self.fp.n, self.T
        pin = lambda *shape, dtype=torch.float64: torch.zeros(shape, dtype=dtype).pin_memory()      # noqa: E731
        h = dict(f=pin(2, T, m), f_old=pin(2, T, m), dq=pin(2, T, n), err=pin(2, T, m), kappa=pin(2, T, m), status=pin(2, T, dtype=torch.int32))
        h['np'] = {k_: v.numpy() for k_, v in h.items()}
        ptr = {k_: [v[i].data_ptr() for i in (0, 1)] for k_, v in h.items() if k_ != 'np'}
        fn = _lib.lib().uvs_rmckf_step_f64
        fpref = C.byref(self.fp)
        # everything of the call that does not change from step to step, per parity
        h['call'] = [lambda first, k, dq_prev, stream, i=i: fn(fpref, self.T, self.X.data_ptr(), self.P.data_ptr(), ptr['f'][i], ptr['f_old'][i],
                                                                dq_prev, first, k, ptr['dq'][i], ptr['err'][i], ptr['kappa'][i],
                                                                ptr['status'][i], stream) for i in (0, 1)]
        h['ptr'], h['calls'] = ptr, 0
        return h

    def step_host(self, f, f_old, k, dq_prev=None):
        """The drop-in step for a caller whose data lives on the HOST (Experiment.run() with an external robot, experiment.py:166-312): ``f``,
        ``f_old`` (T, m) array-likes; ``dq_prev`` (T, n) or None = the command the previous call returned (zero on the first).  No device copy
        in either direction and no allocation: the kernel reads the inputs from and writes its outputs to pinned host memory; one launch, one
        stream synchronisation.  Returns numpy views (dq (T, n), err (T, m), kappa (T, m), status (T,)) that stay valid until the next call
        but one; X and P stay in HBM."""
        h = self._host
        if h is None:
            h = self._host = self._host_io()
        i = h['calls'] & 1
        views = h['np']
        views['f'][i][...] = f
        views['f_old'][i][...] = f_old
        if dq_prev is not None:
            views['dq'][1 - i][...] = dq_prev
        stream = _torch().cuda.current_stream()
        _lib.check(h['call'][i](int(self.first), int(k), h['ptr']['dq'][1 - i], C.c_void_p(stream.cuda_stream)))
        self.first = False
        h['calls'] += 1
        stream.synchronize()
        return views['dq'][i], views['err'][i], views['kappa'][i], views['status'][i]

    def step(self, f, f_old, dq_prev, k):
        """f, f_old: (T, m), dq_prev: (T, n) cuda fp64 tensors.  Updates X, P in place; returns (dq, err, kappa, status)."""
        f, f_old, dq_prev = f.contiguous(), f_old.contiguous(), dq_prev.contiguous()
        rc = _lib.lib().uvs_rmckf_step_f64(C.byref(self.fp), self.T, self.X.data_ptr(), self.P.data_ptr(), f.data_ptr(),
                                           f_old.data_ptr(), dq_prev.data_ptr(), int(self.first), int(k), self.dq.data_ptr(),
                                           self.err.data_ptr(), self.kappa.data_ptr(), self.status.data_ptr(), _stream())
        _lib.check(rc)
        self.first = False
        return self.dq, self.err, self.kappa, self.status


def stats_reduce(err, t, k_done=None, layout='kct'):
    """Per-trial ||ISE||, ||IAE||, ||ITAE|| (results/plot_errorbar.m:39-84) of an error stream tensor."""
    torch = _torch()
    e = as_tkc(err, layout)
    T, K, m = e.shape
    stats = torch.empty((T, 3), dtype=torch.float64, device=err.device)
    t_dev = torch.as_tensor(np.asarray(t, float), device=err.device)
    rc = _lib.lib().uvs_stats_reduce_f64(T, K, m, stream_view(err, layout), t_dev.data_ptr(),
                                         None if k_done is None else k_done.data_ptr(), stats.data_ptr(), _stream())
    _lib.check(rc)
    return stats
