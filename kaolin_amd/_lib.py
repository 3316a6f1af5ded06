"""ctypes binding of libkaolin_amd.so (the C ABI declared in include/kaolin_amd.h).

This is the binding a maintainer of the reference would add in place of
``kaolin/csrc/bindings.cpp:103-115``.  There is NO fallback:
if the shared library is missing or a symbol cannot be resolved the operator
fails loudly -- the product never routes through a CPU path.
"""
import ctypes
import os
import threading

import torch  # noqa: F401  (must be imported first: we share torch's libamdhip64.so.7)

_HERE = os.path.dirname(os.path.abspath(__file__))
# (KAMD_LIB_PATH: development builds of the same library, e.g. the phase-profiling variant `make -C kaolin_amd/csrc prof`)
LIB_PATH = os.environ.get('KAMD_LIB_PATH') or os.path.join(_HERE, 'libkaolin_amd.so')

_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_f = ctypes.c_float
_sz = ctypes.c_size_t
_dbl = ctypes.c_double

# name -> (restype, argtypes).  Lists every symbol include/kaolin_amd.h declares;
# tests/test_abi.py cross-checks this table against the header and the built library.
SIGNATURES = {
    'kamd_version': (ctypes.c_char_p, []),
    'kamd_sided_distance_forward_workspace': (_sz, [_i, _i, _i, _i]),
    'kamd_sided_distance_pair_forward_workspace': (_sz, [_i, _i, _i, _i]),
    'kamd_chamfer_distance_backward_f32': (_i, [_vp, _i, _i, _i, _vp, _f, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'kamd_chamfer_distance_forward_workspace': (_sz, [_i, _i, _i, _i]),
    'kamd_chamfer_distance_forward_f32': (_i, [_vp, _i, _i, _i, _vp, _vp, _f, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'kamd_chamfer_distance_backward_fused_f32': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'kamd_sided_distance_pair_forward_f32': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'kamd_sided_distance_pair_forward_f64': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'kamd_rasterize_forward_workspace': (_sz, [_i, _i, _i, _i64, _i]),
    'kamd_dibr_soft_mask_forward_workspace': (_sz, [_i, _i, _i, _i, _i, _i]),
    'kamd_triangle_distance_forward_workspace': (_sz, [_i, _i, _i]),
    'kamd_dibr_soft_mask_lean_capacity': (_sz, [_i, _i, _i, _i]),
    'kamd_dibr_soft_mask_work_words': (_sz, [_i, _i, _i]),
    'kamd_dibr_rasterization_workspace': (_sz, [_i, _i, _i, _i, _i, _i]),
    'kamd_trianglemeshes_to_voxelgrids_workspace': (_sz, [_i, _i, _i]),
    'kamd_trianglemeshes_to_voxelbits_words': (_sz, [_i]),
    'kamd_mask_iou_workspace': (_sz, [_i]),
    'kamd_weighted_sum2_workspace': (_sz, []),
    'kamd_deftet_forward_workspace': (_sz, [_i, _i, _i, _i]),
    'kamd_mesh_to_spc_stage_levels': (_i, []),
    'kamd_mesh_to_spc_scan_workspace': (_sz, [_i64]),
    'kamd_mesh_to_spc_stage_count': (_i, [_vp, _i64, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    'kamd_mesh_to_spc_stage_emit': (_i, [_vp, _i64, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    'kamd_mesh_to_spc_build_workspace': (_sz, [_i64, _i]),
    'kamd_mesh_to_spc_build': (_i, [_vp, _i64, _i, _vp, _vp, _vp, _sz, _vp]),
    'kamd_mesh_to_spc_results': (_i, [_vp, _i64, _i, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    'kamd_profile_enable': (_i, [_i]),
    'kamd_profile_select': (_i, [_i]),
    'kamd_profile_reset': (_i, []),
    'kamd_profile_num_kernels': (_i, []),
    'kamd_profile_kernel_name': (ctypes.c_char_p, [_i]),
    'kamd_profile_read': (_i, [_i, _vp, _vp]),
    'kamd_triangle_distance_work_counters': (_i, [_i, _vp]),
    'kamd_debug_transpose64': (_i, [_vp, _i, _vp, _vp, _i]),
}
for _t in ('f32', 'f64', 'f16', 'u8', 'i16', 'i32', 'i64'):
    SIGNATURES[f'kamd_sided_distance_forward_{_t}'] = (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp])
    SIGNATURES[f'kamd_sided_distance_backward_{_t}'] = (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp])
for _t in ('f32', 'f64'):
    SIGNATURES[f'kamd_packed_rasterize_forward_{_t}'] = (
        _i, [_vp, _i, _i, _i, _i, _i64, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp])
    SIGNATURES[f'kamd_rasterize_backward_{_t}'] = (
        _i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp])
    SIGNATURES[f'kamd_dibr_soft_mask_forward_{_t}'] = (
        _i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp])
    SIGNATURES[f'kamd_dibr_soft_mask_backward_{_t}'] = (
        _i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp])
    SIGNATURES[f'kamd_dibr_soft_mask_forward_lean_{_t}'] = (
        _i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp])
    SIGNATURES[f'kamd_dibr_soft_mask_backward_lean_{_t}'] = (
        _i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _f, _f, _vp])
    SIGNATURES[f'kamd_rasterize_forward_fused_{_t}'] = (
        _i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _dbl, _f, _vp, _vp, _vp, _vp])
    SIGNATURES[f'kamd_dibr_soft_mask_forward_fused_{_t}'] = (
        _i, [_vp, _i, _i, _i, _i, _i, _vp, _dbl, _dbl, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp])
    SIGNATURES[f'kamd_dibr_rasterization_forward_{_t}'] = (
        _i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _dbl, _f, _f, _dbl] + [_vp] * 11)
    SIGNATURES[f'kamd_rasterize_forward_fused_strided_{_t}'] = (
        _i, [_vp, _i, _i, _i, _i, _i, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _dbl, _f, _vp, _vp, _vp, _vp])
    SIGNATURES[f'kamd_dibr_rasterization_backward_{_t}'] = (
        _i, [_vp, _i, _i, _i, _i, _i, _i] + [_vp] * 12 + [_dbl, _f, _f, _vp, _vp])
    SIGNATURES[f'kamd_mask_iou_forward_{_t}'] = (_i, [_vp, _i, _i64, _vp, _vp, _vp, _vp, _vp])
    SIGNATURES[f'kamd_mask_iou_backward_{_t}'] = (_i, [_vp, _i, _i64, _vp, _vp, _vp, _vp])
    SIGNATURES[f'kamd_weighted_sum2_forward_{_t}'] = (_i, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp])
    SIGNATURES[f'kamd_weighted_sum2_backward_{_t}'] = (_i, [_vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp])
    SIGNATURES[f'kamd_texture_mapping_forward_{_t}'] = (_i, [_vp, _i, _i64, _i, _i, _i, _i, _vp, _vp, _vp])
    SIGNATURES[f'kamd_texture_mapping_backward_{_t}'] = (_i, [_vp, _i, _i64, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp])
    SIGNATURES[f'kamd_prepare_vertices_forward_{_t}'] = (
        _i, [_vp, _i, _i, _i, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp])
    SIGNATURES[f'kamd_prepare_vertices_backward_{_t}'] = (
        _i, [_vp, _i, _i, _i, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp])
    SIGNATURES[f'kamd_mesh_intersection_{_t}'] = (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp])
    SIGNATURES[f'kamd_deftet_sparse_render_forward_{_t}'] = (
        _i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _sz])
    SIGNATURES[f'kamd_deftet_sparse_render_forward_fused_{_t}'] = (
        _i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f] + [_vp] * 9 + [_sz])
    SIGNATURES[f'kamd_deftet_sparse_render_backward_{_t}'] = (
        _i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp])
    SIGNATURES[f'kamd_triangle_distance_forward_{_t}'] = (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp])
    SIGNATURES[f'kamd_triangle_distance_backward_{_t}'] = (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp])
    SIGNATURES[f'kamd_trianglemeshes_to_voxelgrids_{_t}'] = (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp])
    SIGNATURES[f'kamd_trianglemeshes_to_voxelbits_{_t}'] = (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp])

_lock = threading.Lock()
_lib = None


def load():
    """Returns the loaded library (loading it on first use). Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f'kaolin_amd: {LIB_PATH} is missing -- build it with '
                    '`python -c "import __graft_entry__ as g; g.build()"` or `make -C kaolin_amd/csrc`. '
                    'There is no CPU fallback for the HIP operators.')
            lib = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)  # AttributeError if the symbol is not exported
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def check(status, what):
    """Maps a non-zero hipError_t returned by the C ABI to RuntimeError (the reference
    surfaces kernel errors through AT_CUDA_CHECK(cudaGetLastError()))."""
    if status != 0:
        raise RuntimeError(f'HIP error {status} in {what}')


_PRETTY = {torch.float16: 'Half', torch.float32: 'Float', torch.float64: 'Double', torch.int64: 'Long',
           torch.int32: 'Int', torch.int16: 'Short', torch.uint8: 'Byte', torch.bfloat16: 'BFloat16',
           torch.bool: 'Bool', torch.int8: 'Char'}


def dtype_suffix(dtype, what, allowed=('f32', 'f64')):
    s = {torch.float32: 'f32', torch.float64: 'f64', torch.float16: 'f16', torch.uint8: 'u8', torch.int16: 'i16',
         torch.int32: 'i32', torch.int64: 'i64'}.get(dtype)
    if s is None or s not in allowed:
        # reference: AT_ERROR(name, " not implemented for '", toString(TYPE), "'")
        raise RuntimeError(f'"{what}" not implemented for \'{_PRETTY.get(dtype, dtype)}\'')
    return s


def pretty_dtype(dtype):
    return _PRETTY.get(dtype, str(dtype))


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_ptr(device):
    """The current stream of `device` as the C ABI's ``void* stream`` (a hipStream_t).  ``torch.cuda.current_stream(device)``
    builds a Stream object through three layers of Python (~7 us, three times per DIB-R step): the raw handle is one C call."""
    if _raw_stream is not None:
        index = device.index
        return _raw_stream(torch.cuda.current_device() if index is None else index)
    return torch.cuda.current_stream(device).cuda_stream


class on_device:
    """``with torch.cuda.device(device)`` that costs nothing when `device` already is the current one (the usual case:
    the context manager is ~5 us of host time per operator call)."""
    __slots__ = ('_ctx',)

    def __init__(self, device):
        index = device.index
        self._ctx = None if (index is None or index == torch.cuda.current_device()) else torch.cuda.device(device)

    def __enter__(self):
        if self._ctx is not None:
            self._ctx.__enter__()

    def __exit__(self, *exc):
        if self._ctx is not None:
            return self._ctx.__exit__(*exc)
        return False


def ptr(t):
    """Device address for a ``void*`` / ``T*`` parameter (argtypes are declared: a plain int converts; None -> NULL)."""
    return t.data_ptr() if t is not None else None


def workspace(nbytes, device):
    """Scratch for one call, owned by torch's caching allocator (stream-ordered reuse)."""
    if nbytes <= 0:
        return None
    return torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=device)


def kernel_profile(reset=False):
    """{kernel name: (total_ms, launches)} accumulated since the last reset while profiling was enabled
    (kamd_profile_enable).  Synchronises the recorded events."""
    lib = load()
    out = {}
    for k in range(lib.kamd_profile_num_kernels()):
        ms, n = ctypes.c_double(0), ctypes.c_int64(0)
        lib.kamd_profile_read(k, ctypes.byref(ms), ctypes.byref(n))
        if n.value:
            out[lib.kamd_profile_kernel_name(k).decode()] = (ms.value, n.value)
    if reset:
        lib.kamd_profile_reset()
    return out
