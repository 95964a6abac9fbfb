"""ctypes binding of libuvs_rmckf.so (include/uvs_rmckf.h).

The product path has no CPU fallback: if the HIP library is missing or fails to load,
``lib()`` raises ``UvsLibraryError`` -- nothing in this package silently substitutes numpy.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('UVS_LIB_PATH', os.path.join(HERE, 'libuvs_rmckf.so'))   # override only for diagnostic builds
CSRC = os.path.join(HERE, 'csrc')

UVS_MAX_M, UVS_MAX_N, UVS_MAX_POINTS = 32, 8, 16
METHOD_KF, METHOD_MCKF, METHOD_IMCCKF, METHOD_GMCKF = 2, 3, 4, 5
PLANT_DH_PINHOLE, PLANT_LINEAR = 0, 1


class UvsLibraryError(RuntimeError):
    pass


class UvsError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f'libuvs_rmckf error {code}: {text}')
        self.code = code


class View(C.Structure):
    _fields_ = [('base', C.c_void_p), ('trial_stride', C.c_int64), ('step_stride', C.c_int64), ('comp_stride', C.c_int64)]


class FilterParams(C.Structure):
    _fields_ = [('m', C.c_int32), ('n', C.c_int32), ('method', C.c_int32), ('annealing', C.c_int32), ('k_max', C.c_int32),
                ('steps', C.c_int32), ('initial_guess', C.c_int32), ('lanes_per_filter', C.c_int32),
                ('kernel_bw', C.c_double), ('anneal_span', C.c_double), ('gain', C.c_double), ('dt', C.c_double),
                ('reg', C.c_double), ('fpi_threshold', C.c_double), ('fpi_epoch_max', C.c_int32), ('reserved', C.c_int32),
                ('desired', C.c_double * UVS_MAX_M)]


class NoiseParams(C.Structure):
    _fields_ = [('type', C.c_int32), ('m', C.c_int32), ('steps', C.c_int32), ('hold_cnt', C.c_int32)] + \
               [(k, C.c_double) for k in ('std', 'mean', 'rho', 'alpha', 'beta', 'gamma', 'delta', 'inv_alpha', 'expo', 'one_minus_alpha',
                                          'cms_const', 'cms_B', 'cms_S', 'shift', 'sqrt2', 'two_over_pi')]


class Plant(C.Structure):
    _fields_ = [('n_joints', C.c_int32), ('n_points', C.c_int32),
                ('theta_offset', C.c_double * UVS_MAX_N), ('d', C.c_double * UVS_MAX_N), ('a', C.c_double * UVS_MAX_N),
                ('cos_alpha', C.c_double * UVS_MAX_N), ('sin_alpha', C.c_double * UVS_MAX_N),
                ('points', (C.c_double * 3) * UVS_MAX_POINTS), ('focal', C.c_double), ('center', C.c_double),
                ('kind', C.c_int32), ('reserved', C.c_int32),
                ('lin_jacobian', C.c_void_p), ('lin_f0', C.c_void_p), ('lin_q0', C.c_void_p)]


# name -> (restype, argtypes); every symbol include/uvs_rmckf.h declares
_VP, _I64, _I32 = C.c_void_p, C.c_int64, C.c_int32
SYMBOLS = {
    'uvs_version': (C.c_char_p, []),
    'uvs_last_error': (C.c_char_p, []),
    'uvs_supported_lanes': (C.c_int, [_I32, _I32, C.POINTER(_I32), _I32]),
    'uvs_rmckf_closed_loop_f64': (C.c_int, [C.POINTER(FilterParams), C.POINTER(Plant), _I64] + [View] * 8 + [_VP] * 3 + [View] * 2 + [_VP]),
    'uvs_rmckf_closed_loop_ws_f64': (C.c_int, [C.POINTER(FilterParams), C.POINTER(Plant), _I64] + [View] * 8 + [_VP] * 3 + [View] * 2 + [_VP, C.c_size_t, _VP]),
    'uvs_rmckf_closed_loop_lanes': (C.c_int, [C.POINTER(FilterParams), C.POINTER(Plant), _I64]),
    'uvs_rmckf_closed_loop_segments': (C.c_int, [C.POINTER(FilterParams), C.POINTER(Plant), _I64]),
    'uvs_rmckf_closed_loop_workspace_bytes': (C.c_size_t, [C.POINTER(FilterParams), C.POINTER(Plant), _I64]),
    'uvs_rmckf_closed_loop_fallback_offset': (C.c_size_t, [C.POINTER(FilterParams), C.POINTER(Plant), _I64]),
    'uvs_rmckf_replay_f64': (C.c_int, [C.POINTER(FilterParams), _I64] + [View] * 7 + [_VP] * 2 + [View] * 2 + [_VP]),
    'uvs_rmckf_replay_f32': (C.c_int, [C.POINTER(FilterParams), _I64] + [View] * 5 + [_VP] * 2 + [_VP]),
    'uvs_rmckf_step_f64': (C.c_int, [C.POINTER(FilterParams), _I64] + [_VP] * 5 + [_I32, _I32] + [_VP] * 4 + [_VP]),
    'uvs_stats_reduce_f64': (C.c_int, [_I64, _I32, _I32, View, _VP, _VP, _VP, _VP]),
    'uvs_debug_math_f64': (C.c_int, [_I32, _I64, _VP, _VP, _VP]),
    'uvs_noise_generate_f64': (C.c_int, [C.POINTER(NoiseParams), _I64, _VP, _VP, View, _VP]),
    'uvs_noise_generate_streams_f64': (C.c_int, [C.POINTER(NoiseParams), _I64, _VP, _VP, _VP, _I64, _I64, _VP]),
    'uvs_noise_kernel_variant': (C.c_int, [C.POINTER(NoiseParams)]),
    'uvs_pcg64_seed_u64': (C.c_int, [_I64, _VP, _VP, _VP]),
}

_lib = None


def build(force=False):
    """Compile libuvs_rmckf.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.run(['make', '-j', str(os.cpu_count() or 4), '-C', CSRC], check=True)      # one translation unit per kernel family
    return LIB_PATH


def lib():
    """The loaded library with typed entry points; raises UvsLibraryError when it cannot be loaded."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise UvsLibraryError(f'{LIB_PATH} is missing: run `make -C {CSRC}` (or __graft_entry__.build()); '
                                  'there is no CPU fallback for the RMCKF path')
        try:
            # One HIP runtime per process: PyTorch bundles its own libamdhip64 (soname libamdhip64.so.7).  Loading it first makes
            # our DT_NEEDED libamdhip64.so.7 resolve to that same instance, so torch tensors, streams and our kernels share one
            # runtime; the other order loads /opt/rocm's copy as a second runtime that sees no device.
            import torch  # noqa: F401
            handle = C.CDLL(LIB_PATH)
        except OSError as exc:
            raise UvsLibraryError(f'cannot load {LIB_PATH}: {exc}') from exc
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(code):
    if code != 0:
        raise UvsError(code, lib().uvs_last_error().decode())


def view_of(tensor, dims):
    """uvs_view over a torch tensor.  ``dims`` names the tensor axes that play (trial, step, comp), e.g. a
    [step][comp][trial] tensor is view_of(t, (2, 0, 1)); use None for an absent axis (stride 0)."""
    if tensor is None:
        return View(None, 0, 0, 0)
    assert tensor.dtype.is_floating_point and tensor.element_size() in (4, 8), 'fp64 tensors (fp32 for uvs_rmckf_replay_f32)'
    strides = [0 if d is None else tensor.stride(d) for d in dims]
    return View(tensor.data_ptr(), *strides)


NULL_VIEW = View(None, 0, 0, 0)
