"""ATen-style argument checks with the reference's exact error strings.

The reference's .cpp wrappers validate arguments with at::checkSize /
checkAllSameGPU / checkAllContiguous / checkSameType (e.g.
kaolin/csrc/metrics/sided_distance.cpp:68-78) and its tests regex-match the
resulting messages (tests/python/kaolin/metrics/test_pointcloud.py:126-151).
These helpers reproduce ATen/TensorUtils.cpp's formatting.
"""
import torch


class Arg:
    """at::TensorArg{tensor, name, pos}"""
    __slots__ = ('t', 'name', 'pos')

    def __init__(self, t, name, pos):
        self.t, self.name, self.pos = t, name, pos

    def __str__(self):
        return f"argument #{self.pos} '{self.name}'"


def check_same_gpu(fn, a, b):
    a_cpu, b_cpu = not a.t.is_cuda, not b.t.is_cuda
    if a_cpu or b_cpu:
        msg = ''
        if a_cpu:
            msg += f'Tensor for {a} is on CPU, '
        if b_cpu:
            msg += f'Tensor for {b} is on CPU, '
        msg += ('but expected ' + ('them' if (not a_cpu and not b_cpu) else 'it') +
                f' to be on GPU (while checking arguments for {fn})')
        raise RuntimeError(msg)
    if a.t.get_device() != b.t.get_device():
        raise RuntimeError(
            f'Expected tensor for {a} to have the same device as tensor for {b}; but device '
            f'{a.t.get_device()} does not equal {b.t.get_device()} (while checking arguments for {fn})')


def check_all_same_gpu(fn, args):
    args = [a for a in args if a.t is not None]
    if len(args) == 1:
        check_same_gpu(fn, args[0], args[0])
    for a in args[1:]:
        check_same_gpu(fn, args[0], a)


def check_all_contiguous(fn, args):
    for a in args:
        if a.t is not None and not a.t.is_contiguous():
            raise RuntimeError(
                f'Expected contiguous tensor, but got non-contiguous tensor for {a} '
                f'(while checking arguments for {fn})')


def check_same_type(fn, a, b):
    if a.t.dtype != b.t.dtype:
        raise RuntimeError(
            f'Expected tensor for {a} to have the same type as tensor for {b}; but type '
            f'{a.t.type()} does not equal {b.t.type()} (while checking arguments for {fn})')


def check_all_same_type(fn, args):
    for a in args[1:]:
        check_same_type(fn, args[0], a)


def check_dim(fn, a, dim):
    if a.t.dim() != dim:
        raise RuntimeError(
            f'Expected {dim}-dimensional tensor, but got {a.t.dim()}-dimensional tensor for {a} '
            f'(while checking arguments for {fn})')


def check_size(fn, a, sizes):
    sizes = [int(s) for s in sizes]
    check_dim(fn, a, len(sizes))
    if list(a.t.shape) != sizes:
        raise RuntimeError(
            f'Expected tensor of size {sizes}, but got tensor of size {list(a.t.shape)} for {a} '
            f'(while checking arguments for {fn})')


def check_same_size(fn, a, b):
    if list(a.t.shape) != list(b.t.shape):
        raise RuntimeError(
            f'Expected tensor for {a} to have same size as tensor for {b}; but {list(a.t.shape)} '
            f'does not equal {list(b.t.shape)} (while checking arguments for {fn})')


def torch_check(cond, msg):
    """TORCH_CHECK(cond, msg)"""
    if not cond:
        raise RuntimeError(msg)
