"""``kaolin_amd._C`` -- the operator namespace that mirrors the reference's pybind11
module ``kaolin._C`` for the DIB-R / 3D-metrics hot path
(kaolin/csrc/bindings.cpp:103-115):

    _C.render.mesh.packed_rasterize_forward_cuda / rasterize_backward_cuda
    _C.render.mesh.dibr_soft_mask_forward_cuda   / dibr_soft_mask_backward_cuda
    _C.metrics.sided_distance_forward_cuda       / sided_distance_backward_cuda
    _C.metrics.unbatched_triangle_distance_forward_cuda / _backward_cuda

Same names, argument order, allocation/ownership rules and error strings; the work
is done by hand-written HIP kernels in libkaolin_amd.so through the C ABI of
include/kaolin_amd.h (the ``_cuda`` suffix is kept because callers spell it that
way; on ROCm ``tensor.is_cuda`` is the HIP device).
"""
from . import metrics, render, ops  # noqa: F401
