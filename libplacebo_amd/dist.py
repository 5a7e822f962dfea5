"""Multi-GPU helpers (one process per GPU, torch.distributed; "nccl" is RCCL on ROCm).

Streams are independent: there is no collective on the data path. The one optional exchange is
the peak-detection measurement when several ranks render tiles or frames of the SAME scene and
must tone-map with one common peak: an all-reduce of the 816-word buffer,
SUM on every field except frame_max_pq (words 36..47), which takes MAX.
"""

PEAK_WORDS = 816
MAX_LO, MAX_HI = 36, 48


def allreduce_peak_buffer(buf, dist):
    """In-place all-reduce of a peak buffer held in an integer torch tensor of 816 elements
    (int32 on the GPU for RCCL, any integer dtype for gloo)."""
    assert buf.numel() == PEAK_WORDS
    mx = buf[MAX_LO:MAX_HI].clone()
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    buf[MAX_LO:MAX_HI] = mx
    return buf


def allreduce_renderer_peak(renderer, dist):
    """All-reduce the pending measurement of a pl_renderer's tone-mapping state across ranks.
    Call between the detection pass and the frame that consumes it (e.g. with
    `allow_delayed = true`, between two pl_render_image calls). No-op if nothing is pending."""
    import ctypes as C
    import torch
    from . import lib
    L = lib()
    L.pl_hip_peak_buffer.restype = C.c_void_p
    L.pl_hip_peak_buffer.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    state = L.pl_hip_renderer_tone_map_state(renderer.rr)
    size = C.c_size_t()
    ptr = L.pl_hip_peak_buffer(state, C.byref(size)) if state else None
    if not ptr:
        return False
    # wrap the device buffer without copying it
    class _Arr:
        __cuda_array_interface__ = {"shape": (PEAK_WORDS,), "typestr": "<i4",
                                    "data": (ptr, False), "version": 2}
    torch.cuda.synchronize()
    buf = torch.as_tensor(_Arr(), device="cuda")
    allreduce_peak_buffer(buf, dist)
    torch.cuda.synchronize()
    return True
