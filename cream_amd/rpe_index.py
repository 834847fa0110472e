"""rpe_index operator — host-side mirror of the reference's `rpe_index_cpp` module and
`RPEIndexFunction` (iRPE/DeiT-with-iRPE/rpe_ops/rpe_index.cpp:130-141,
rpe_ops/rpe_index.py:11-56) on top of the C ABI (include/cream_amd.h).

    Y[b, h, i, j]       = input[b, h, i, index[i, j]]                 (forward)
    gin[b, h, i, u]    += sum_{j: index[i, j] == u} gout[b, h, i, j]  (backward)

Argument meaning, ownership and error behaviour follow the reference: outputs are
allocated here, `grad_input` is allocated/zeroed by the caller and updated in place,
shape/dtype/device violations raise RuntimeError with the reference's messages.  Device
tensors always run the HIP kernels on the current stream (no sync, no fallback); the
`*_cpu` functions are the reference's own CPU entry points for host tensors.
bf16 is supported in addition to the reference's float/double/half.
"""
import torch

from . import _lib, timing

_DTYPES = {torch.float32: _lib.F32, torch.float16: _lib.F16,
           torch.bfloat16: _lib.BF16, torch.float64: _lib.F64}


def version():
    """rpe_index.cpp:126-128 — asserted == "1.2.0" by rpe_ops/rpe_index.py:5-8."""
    return _lib.version()


def _assert(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _dtype_code(t, opname):
    code = _DTYPES.get(t.dtype)
    if code is None:
        raise RuntimeError(f'"{opname}" not implemented for \'{str(t.dtype).replace("torch.", "")}\'')
    return code


def _check_fwd(input, index, gpu):
    dev = "GPU" if gpu else "CPU"
    is_dev = (lambda t: t.is_cuda) if gpu else (lambda t: t.device.type == "cpu")
    _assert(is_dev(input), f"input must be a {dev} tensor")
    _assert(is_dev(index), f"index must be a {dev} tensor")
    _assert(input.dim() == 4, "input must be a 4D tensor")
    _assert(index.dim() == 2, "index must be a 2D tensor")
    _assert(index.dtype == torch.int32, "index must be Int type")


def _check_bwd(grad_input, grad_output, index, gpu):
    dev = "GPU" if gpu else "CPU"
    is_dev = (lambda t: t.is_cuda) if gpu else (lambda t: t.device.type == "cpu")
    _assert(is_dev(grad_input), f"grad_input must be a {dev} tensor")
    _assert(is_dev(grad_output), f"grad_output must be a {dev} tensor")
    _assert(is_dev(index), f"grad_index must be a {dev} tensor")
    _assert(grad_input.dim() == 4, "input must be a 4D tensor")
    _assert(grad_output.dim() == 4, "input must be a 4D tensor")
    _assert(index.dim() == 2, "index must be a 2D tensor")
    _assert(index.dtype == torch.int32, "index must be Int type")
    _assert(grad_input.dtype == grad_output.dtype, "grad_input and grad_output must share a dtype")
    # The reference calls .contiguous() on grad_input and would silently write into a
    # temporary copy; in-place semantics only make sense for a contiguous tensor.
    _assert(grad_input.is_contiguous(), "grad_input must be contiguous (it is updated in place)")


def forward_gpu(input, index):
    """rpe_index_forward_gpu (rpe_index_cuda.cu:54-94): honours arbitrary input strides,
    requires a contiguous index, launches on the current stream, never synchronises."""
    _check_fwd(input, index, gpu=True)
    _assert(index.is_contiguous(), "index should be contiguous")
    _assert(index.device == input.device, "input and index must be on the same device")
    code = _dtype_code(input, "rpe_index_forward_gpu")
    B, H, nb = input.size(0), input.size(1), input.size(3)
    Lq, Lk = index.size(0), index.size(1)
    y = torch.empty((B, H, Lq, Lk), dtype=input.dtype, device=input.device)
    s0, s1, s2, s3 = input.stride()
    es = input.element_size()
    nbytes = B * H * Lq * (nb + Lk) * es + Lq * Lk * 4       # SURVEY §8(d) algorithmic bytes
    with torch.cuda.device(input.device), timing.region("rpe_index_fwd", nbytes):
        stream = torch.cuda.current_stream().cuda_stream
        rc = _lib.load().cream_rpe_index_fwd(y.data_ptr(), input.data_ptr(), index.data_ptr(),
                                             B, H, Lq, Lk, nb, s0, s1, s2, s3, code, stream)
    _lib.check(rc, "cream_rpe_index_fwd")
    return y


def backward_gpu(grad_input, grad_output, index, accumulate=True):
    """rpe_index_backward_gpu (rpe_index_cuda.cu:96-140): accumulates INTO grad_input.
    `accumulate=False` (extension) overwrites instead, for callers that own the buffer."""
    _check_bwd(grad_input, grad_output, index, gpu=True)
    code = _dtype_code(grad_output, "rpe_index_backward_gpu")
    nb = grad_input.size(3)
    B, H, Lq, Lk = grad_output.shape
    _assert(tuple(grad_input.shape[:3]) == (B, H, Lq), "grad_input / grad_output shape mismatch")
    _assert(tuple(index.shape) == (Lq, Lk), "index shape mismatch")
    gout = grad_output.contiguous()
    idx = index.contiguous()
    es = gout.element_size()
    nbytes = B * H * Lq * (nb + Lk) * es + Lq * Lk * 4
    with torch.cuda.device(gout.device), timing.region("rpe_index_bwd", nbytes):
        stream = torch.cuda.current_stream().cuda_stream
        rc = _lib.load().cream_rpe_index_bwd(grad_input.data_ptr(), gout.data_ptr(), idx.data_ptr(),
                                             B, H, Lq, Lk, nb, code, int(bool(accumulate)), stream)
    _lib.check(rc, "cream_rpe_index_bwd")


def forward_cpu(input, index):
    """rpe_index_forward_cpu (rpe_index.cpp:8-73) for HOST tensors."""
    _check_fwd(input, index, gpu=False)
    code = _dtype_code(input, "rpe_index_forward_cpu")
    B, H, nb = input.size(0), input.size(1), input.size(3)
    Lq, Lk = index.size(0), index.size(1)
    y = torch.empty((B, H, Lq, Lk), dtype=input.dtype)
    inp, idx = input.contiguous(), index.contiguous()
    rc = _lib.load().cream_rpe_index_fwd_host(y.data_ptr(), inp.data_ptr(), idx.data_ptr(),
                                              B, H, Lq, Lk, nb, code)
    _lib.check(rc, "cream_rpe_index_fwd_host")
    return y


def backward_cpu(grad_input, grad_output, index):
    """rpe_index_backward_cpu (rpe_index.cpp:82-124) for HOST tensors."""
    _check_bwd(grad_input, grad_output, index, gpu=False)
    code = _dtype_code(grad_input, "rpe_index_backward_atomic_cpu")
    nb = grad_input.size(3)
    Lq, Lk = index.size(0), index.size(1)
    B, H = grad_output.size(0), grad_output.size(1)
    gout, idx = grad_output.contiguous(), index.contiguous()
    rc = _lib.load().cream_rpe_index_bwd_host(grad_input.data_ptr(), gout.data_ptr(), idx.data_ptr(),
                                              B, H, Lq, Lk, nb, code)
    _lib.check(rc, "cream_rpe_index_bwd_host")


class RPEIndexFunction(torch.autograd.Function):
    """Y[b, h, i, j] = input[b, h, i, index[i, j]]  (rpe_ops/rpe_index.py:11-56)."""

    @staticmethod
    def forward(ctx, input, index):
        ctx.save_for_backward(index)
        ctx.input_shape = input.shape
        fn = forward_cpu if input.device.type == "cpu" else forward_gpu
        return fn(input, index)

    @staticmethod
    def backward(ctx, grad_output):
        index = ctx.saved_tensors[0]
        if ctx.needs_input_grad[0]:
            if grad_output.device.type == "cpu":
                grad_input = grad_output.new_zeros(ctx.input_shape)
                backward_cpu(grad_input, grad_output, index)
            else:
                # same result as new_zeros + accumulate (rpe_ops/rpe_index.py:51-54)
                # without the memset and the re-read of grad_input
                grad_input = grad_output.new_empty(ctx.input_shape)
                backward_gpu(grad_input, grad_output, index, accumulate=False)
            return grad_input, None
        return None, None
