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


# ---- the C-level exchange (include/libplacebo/hip.h) over a communicator of our own ----------
import ctypes as _C
import glob as _glob
import os as _os


def _load_rccl():
    """The RCCL that belongs to the HIP runtime libplacebo_hip.so is linked against (ROCm's).
    NOT the copy torch bundles: that one is bound to torch's own bundled HIP runtime, whose
    streams are not interchangeable with ours; a communicator must be created and used by one
    and the same RCCL instance."""
    cands = ["/opt/rocm/lib/librccl.so", "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"]
    for c in cands:
        try:
            return _C.CDLL(c, mode=_os.RTLD_GLOBAL)
        except OSError:
            continue
    raise OSError("librccl.so not found")


class _UniqueId(_C.Structure):
    _fields_ = [("internal", _C.c_char * 128)]


def rccl_unique_id():
    """ncclGetUniqueId -> bytes (rank 0 creates it, every rank must receive the same one)"""
    uid = _UniqueId()
    rc = _load_rccl().ncclGetUniqueId(_C.byref(uid))
    assert rc == 0, f"ncclGetUniqueId: {rc}"
    return bytes(uid)


class RcclPeakExchange:
    """ncclCommInitRank + pl_hip_rccl_create + pl_hip_set_peak_exchange on one HipGpu: from then
    on every HDR peak measurement made on that GPU is all-reduced across the communicator
    before it is consumed (same-frame or delayed)."""

    def __init__(self, gpu, rank, world, unique_id, device=None):
        from . import lib
        self.gpu, self.L, self.rccl = gpu, lib(), _load_rccl()
        if device is not None:
            hip = _C.CDLL("libamdhip64.so")
            assert hip.hipSetDevice(int(device)) == 0
        uid = _UniqueId.from_buffer_copy(unique_id)
        self.comm = _C.c_void_p()
        self.rccl.ncclCommInitRank.argtypes = [_C.POINTER(_C.c_void_p), _C.c_int, _UniqueId, _C.c_int]
        rc = self.rccl.ncclCommInitRank(_C.byref(self.comm), int(world), uid, int(rank))
        assert rc == 0, f"ncclCommInitRank: {rc}"
        self.L.pl_hip_rccl_create.restype = _C.c_void_p
        self.L.pl_hip_rccl_create.argtypes = [_C.c_void_p, _C.c_void_p, _C.c_void_p]
        # the all-reduce entry of the very RCCL instance that owns the communicator
        fn_ar = _C.cast(self.rccl.ncclAllReduce, _C.c_void_p)
        self.x = _C.c_void_p(self.L.pl_hip_rccl_create(gpu.gpu, self.comm, fn_ar))
        assert self.x, gpu.messages[-3:]
        fn = _C.cast(self.L.pl_hip_rccl_peak_exchange, _C.c_void_p)
        self.L.pl_hip_set_peak_exchange.argtypes = [_C.c_void_p, _C.c_void_p, _C.c_void_p]
        self.L.pl_hip_set_peak_exchange(gpu.gpu, fn, self.x)

    def stats(self):
        err = _C.c_int()
        self.L.pl_hip_rccl_stats.argtypes = [_C.c_void_p, _C.POINTER(_C.c_int)]
        n = self.L.pl_hip_rccl_stats(self.x, _C.byref(err))
        return n, err.value

    def close(self):
        self.L.pl_hip_set_peak_exchange(self.gpu.gpu, None, None)
        self.gpu.finish()
        self.L.pl_hip_rccl_destroy.argtypes = [_C.POINTER(_C.c_void_p)]
        self.L.pl_hip_rccl_destroy(_C.byref(self.x))
        self.rccl.ncclCommDestroy.argtypes = [_C.c_void_p]
        self.rccl.ncclCommDestroy(self.comm)


# ---- the same exchange over a host-side transport (gloo, MPI, a socket ...) -----------------------
def _hip_runtime(L):
    """hipStreamSynchronize / hipMemcpy of the HIP runtime libplacebo_hip.so itself is linked
    against: looked up THROUGH the library's handle (dlsym on a handle searches the object and its
    dependency tree), never by path -- another package in the process may have mapped a second
    runtime, whose streams are not ours."""
    sync, cpy = L.hipStreamSynchronize, L.hipMemcpy
    sync.argtypes, sync.restype = [_C.c_void_p], _C.c_int
    cpy.argtypes, cpy.restype = [_C.c_void_p, _C.c_void_p, _C.c_size_t, _C.c_int], _C.c_int
    return sync, cpy


class HostPeakExchange:
    """pl_hip_set_peak_exchange with the reduction done on the host: the 816-word measurement is
    read back when the renderer is about to consume it, handed to `reduce(words)` -- a callable
    that all-reduces an int64 numpy array of 816 words in place across the ranks (SUM everywhere,
    MAX on words 36..47; `gloo_reduce(dist)` below) -- and written back. For ranks that share no
    xGMI / RCCL communicator (several processes on one device, several hosts): correctness is the
    same as the RCCL entry, the cost is one stream sync + 6.5 KB over PCIe per measured frame."""

    def __init__(self, gpu, reduce):
        import numpy as np
        from . import lib
        self.gpu, self.L, self.calls, self.errors, self.last_error = gpu, lib(), 0, 0, None
        sync, cpy = _hip_runtime(self.L)
        np_ = np

        # An exception raised inside a ctypes callback is printed and swallowed: every failure is
        # caught here instead and counted (`errors`, `last_error`), and the buffer is then left as
        # it was -- the rank goes on with its LOCAL measurement, like pl_hip_rccl_peak_exchange
        # after a failed RCCL step. (A rank that fails BEFORE `reduce` does not enter the
        # collective: give the process group a timeout if the other ranks must not wait forever.)
        @_C.CFUNCTYPE(None, _C.c_void_p, _C.c_void_p, _C.c_size_t, _C.c_void_p)
        def exchange(priv, words, size, stream):
            try:
                if size != PEAK_WORDS * 4:
                    raise RuntimeError(f"peak buffer of {size} bytes")
                rc = sync(stream)
                if rc:
                    raise RuntimeError(f"hipStreamSynchronize: {rc}")
                buf = np_.zeros(PEAK_WORDS, np_.uint32)
                rc = cpy(buf.ctypes.data, words, size, 2)      # hipMemcpyDeviceToHost
                if rc:
                    raise RuntimeError(f"hipMemcpy (device to host): {rc}")
                wide = buf.astype(np_.int64)
                reduce(wide)
                buf[:] = wide.astype(np_.uint32)
                rc = cpy(words, buf.ctypes.data, size, 1)      # hipMemcpyHostToDevice
                if rc:
                    raise RuntimeError(f"hipMemcpy (host to device): {rc}")
                self.calls += 1
            except BaseException as e:      # noqa: BLE001 (nothing may leave the trampoline)
                self.errors += 1
                self.last_error = repr(e)

        self._cb = exchange     # (keep the trampoline alive)
        self.L.pl_hip_set_peak_exchange.argtypes = [_C.c_void_p, _C.c_void_p, _C.c_void_p]
        self.L.pl_hip_set_peak_exchange(gpu.gpu, _C.cast(exchange, _C.c_void_p), None)

    def close(self):
        self.L.pl_hip_set_peak_exchange(self.gpu.gpu, None, None)


def gloo_reduce(dist):
    """reduce(words) for HostPeakExchange over a torch.distributed process group (any backend that
    takes CPU tensors)"""
    import torch

    def reduce(words):
        t = torch.from_numpy(words)
        allreduce_peak_buffer(t, dist)
    return reduce
