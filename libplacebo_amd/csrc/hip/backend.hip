/*
 * libplacebo-hip — HIP runtime glue (device, memory, copies, events).
 * Plays the role of the driver calls inside a reference backend such as
 * src/opengl/gpu.c or src/vulkan/gpu.c; the pl_gpu-level semantics live in
 * csrc/host/gpu.c.
 */
#include <hip/hip_runtime.h>
#include <string.h>

#include "backend.h"
#include "devmath.hiph"

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return -(int) e_; } while (0)

extern "C" {

int plh_dev_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

const char *plh_strerror(int err)
{
    if (err == -1000)
        return "LDS tile does not fit";
    if (err == -1003)
        return "frame mixing ops need a plain (nearest / bilinear) sampler";
    if (err == -1005)
        return "could not prepare the white-noise dither plane";
    if (err == -1004)
        return "lut3d_tricubic: the colour map must be a pass of its own (plain sampler, no "
               "peak detection / mixing in the same shader)";
    return hipGetErrorString((hipError_t) (err < 0 ? -err : err));
}

int plh_dev_open(int device, struct plh_dev_info *info)
{
    CHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, device));
    if (info) {
        memset(info, 0, sizeof(*info));
        strncpy(info->name, prop.name, sizeof(info->name) - 1);
        strncpy(info->arch, prop.gcnArchName, sizeof(info->arch) - 1);
        info->compute_units = prop.multiProcessorCount;
        info->wavefront_size = prop.warpSize;
        info->lds_per_block = prop.sharedMemPerBlock;
        info->total_mem = prop.totalGlobalMem;
        info->clock_khz = prop.clockRate;
        memcpy(info->uuid, prop.uuid.bytes, 16);
        info->pci_domain = prop.pciDomainID;
        info->pci_bus = prop.pciBusID;
        info->pci_device = prop.pciDeviceID;
    }
    return 0;
}

int plh_stream_create(int device, plh_stream *out)
{
    CHK(hipSetDevice(device));
    hipStream_t s;
    CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = (plh_stream) s;
    return 0;
}

void plh_stream_destroy(plh_stream s)
{
    if (s)
        (void) hipStreamDestroy((hipStream_t) s);
}

int plh_stream_sync(plh_stream s)
{
    CHK(hipStreamSynchronize((hipStream_t) s));
    return 0;
}

// 1 = everything submitted so far has completed, 0 = still running
int plh_stream_idle(plh_stream s)
{
    const hipError_t r = hipStreamQuery((hipStream_t) s);
    if (r == hipSuccess)
        return 1;
    if (r == hipErrorNotReady)
        return 0;
    return -(int) r;
}

void *plh_malloc(int device, size_t size)
{
    void *p = NULL;
    if (hipSetDevice(device) != hipSuccess)
        return NULL;
    if (hipMalloc(&p, size) != hipSuccess)
        return NULL;
    return p;
}

void plh_free(void *ptr)
{
    if (ptr)
        (void) hipFree(ptr);
}

void *plh_host_alloc(size_t size)
{
    void *p = NULL;
    if (hipHostMalloc(&p, size, hipHostMallocDefault) != hipSuccess)
        return NULL;
    return p;
}

// fine-grained (host-coherent) pinned memory, mapped for the device: a kernel's system-scope
// stores become visible to a polling host thread while the stream keeps running
void *plh_host_alloc_coherent(size_t size)
{
    void *p = NULL;
    if (hipHostMalloc(&p, size, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess)
        return NULL;
    return p;
}

void plh_host_free(void *ptr)
{
    if (ptr)
        (void) hipHostFree(ptr);
}

int plh_copy2d_h2d(plh_stream s, void *dst, size_t dpitch, const void *src, size_t spitch,
                   size_t row_bytes, size_t rows)
{
    CHK(hipMemcpy2DAsync(dst, dpitch, src, spitch, row_bytes, rows, hipMemcpyHostToDevice,
                         (hipStream_t) s));
    return 0;
}

int plh_copy2d_d2h(plh_stream s, void *dst, size_t dpitch, const void *src, size_t spitch,
                   size_t row_bytes, size_t rows)
{
    CHK(hipMemcpy2DAsync(dst, dpitch, src, spitch, row_bytes, rows, hipMemcpyDeviceToHost,
                         (hipStream_t) s));
    return 0;
}

int plh_copy2d_d2d(plh_stream s, void *dst, size_t dpitch, const void *src, size_t spitch,
                   size_t row_bytes, size_t rows)
{
    CHK(hipMemcpy2DAsync(dst, dpitch, src, spitch, row_bytes, rows, hipMemcpyDeviceToDevice,
                         (hipStream_t) s));
    return 0;
}

int plh_memset(plh_stream s, void *dst, int value, size_t size)
{
    CHK(hipMemsetAsync(dst, value, size, (hipStream_t) s));
    return 0;
}

int plh_event_create(plh_event *out)
{
    hipEvent_t e;
    CHK(hipEventCreate(&e));
    *out = (plh_event) e;
    return 0;
}

void plh_event_destroy(plh_event e)
{
    if (e)
        (void) hipEventDestroy((hipEvent_t) e);
}

int plh_event_record(plh_event e, plh_stream s)
{
    CHK(hipEventRecord((hipEvent_t) e, (hipStream_t) s));
    return 0;
}

// everything submitted to `s` after this call runs after `e` has completed
int plh_stream_wait_event(plh_stream s, plh_event e)
{
    CHK(hipStreamWaitEvent((hipStream_t) s, (hipEvent_t) e, 0));
    return 0;
}

int plh_event_query(plh_event e)
{
    const hipError_t r = hipEventQuery((hipEvent_t) e);
    if (r == hipSuccess)
        return 1;
    if (r == hipErrorNotReady)
        return 0;
    return -(int) r;
}

int plh_event_sync(plh_event e)
{
    CHK(hipEventSynchronize((hipEvent_t) e));
    return 0;
}

int plh_event_elapsed_ns(plh_event a, plh_event b, uint64_t *ns)
{
    float ms = 0.0f;
    CHK(hipEventElapsedTime(&ms, (hipEvent_t) a, (hipEvent_t) b));
    *ns = (uint64_t) ((double) ms * 1e6);
    return 0;
}

} // extern "C"

__global__ void k_clear(const plh_view dst, float4_t color)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x < dst.w && y < dst.h)
        plh_store(dst, x, y, color);
}

extern "C" int plh_launch_clear(plh_stream s, const struct plh_view *dst, const float color[4])
{
    const dim3 block(64, 4), grid((dst->w + 63) / 64, (dst->h + 3) / 4);
    float4_t c = { color[0], color[1], color[2], color[3] };
    hipLaunchKernelGGL(k_clear, grid, block, 0, (hipStream_t) s, *dst, c);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : -(int) err;
}
