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

// ---- what a launcher needs to know about the device that owns its stream -------------------
// The calling thread's current device is not necessarily that device (a host thread may drive
// several pl_gpu objects; hipSetDevice is only called where memory and streams are created), so
// launchers ask the stream. Cached per device: the CU count, and -- per kernel -- whether the
// dynamic-LDS limit has been raised there (hipFuncSetAttribute acts on the current device and
// costs a runtime call: once per kernel and device, not once per launch).
#define PLH_MAX_DEVICES 64
static int g_dev_cus[PLH_MAX_DEVICES];

int plh_stream_device(plh_stream s, int *cus)
{
    hipDevice_t dev = 0;
    if (!s || hipStreamGetDevice((hipStream_t) s, &dev) != hipSuccess)
        (void) hipGetDevice(&dev);      // (the null stream: the current device's)
    if (dev < 0 || dev >= PLH_MAX_DEVICES)
        dev = 0;
    if (cus) {
        int n = __atomic_load_n(&g_dev_cus[dev], __ATOMIC_RELAXED);
        if (n <= 0) {
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
                n = 256;
            __atomic_store_n(&g_dev_cus[dev], n, __ATOMIC_RELAXED);
        }
        *cus = n;
    }
    return dev;
}

// Raise `kernel`'s dynamic-LDS limit to `bytes` on the device that owns `s`; `done` is the
// caller's per-kernel bit mask of devices already served (a static uint64_t next to the launch).
int plh_kernel_needs_lds(const void *kernel, plh_stream s, size_t bytes, uint64_t *done)
{
    const int dev = plh_stream_device(s, NULL);
    const uint64_t bit = 1ull << dev;
    if (__atomic_load_n(done, __ATOMIC_ACQUIRE) & bit)
        return 0;
    int cur = dev;
    (void) hipGetDevice(&cur);
    if (cur != dev && hipSetDevice(dev) != hipSuccess)
        return -(int) hipErrorInvalidDevice;
    const hipError_t err = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes);
    if (cur != dev)
        (void) hipSetDevice(cur);
    if (err != hipSuccess)
        return -(int) err;
    __atomic_fetch_or(done, bit, __ATOMIC_RELEASE);
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

// A stream whose kernels may only run on `ncus` of the device's compute units (0 or >= all of
// them: an ordinary stream). The driver deals the bits of a CU mask out to the XCDs in turn -- bit
// k is a CU of XCD k % 8 -- so the low `ncus` bits are ncus / 8 CUs on every XCD: the measuring
// pass keeps every L2 and every memory channel, but its waves interleave with the scaler's on that
// many CUs only instead of on all 256.
int plh_stream_create_masked(int device, int ncus, plh_stream *out)
{
    CHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, device));
    if (ncus <= 0 || ncus >= prop.multiProcessorCount)
        return plh_stream_create(device, out);
    uint32_t mask[32] = {0};
    const int words = (prop.multiProcessorCount + 31) / 32;
    if (words > 32)
        return plh_stream_create(device, out);
    for (int k = 0; k < ncus; k++)
        mask[k / 32] |= 1u << (k % 32);
    hipStream_t s;
    CHK(hipExtStreamCreateWithCUMask(&s, (uint32_t) words, mask));
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

// ---- an event offered for the end of the next pass (devmath.hiph: PLH_LAUNCH_LAST) ---------------
static thread_local hipEvent_t g_stop_offer = nullptr;
static thread_local bool g_stop_taken = false;

void plh_launch_offer_stop(plh_event e)
{
    g_stop_offer = (hipEvent_t) e;
    g_stop_taken = false;
}

// 1 = the pass's last kernel carried the event; either way the offer is withdrawn
int plh_launch_stop_taken(void)
{
    const bool taken = g_stop_taken;
    g_stop_offer = nullptr;
    g_stop_taken = false;
    return taken;
}

hipEvent_t plh_take_stop_event(void)
{
    hipEvent_t e = g_stop_offer;
    if (e) {
        g_stop_offer = nullptr;
        g_stop_taken = true;
    }
    return e;
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

// pl_buf_copy_swap (reference src/gpu/utils.c:1065-1138: a GLSL compute pass through the GPU's own
// dispatch, one invocation per 32-bit word): the bytes of every 16-bit half (wordsize 2) or of the
// whole word (wordsize 4) reversed; src == dst (same offset) is an in-place swap.
__global__ void k_swap_words(const uint32_t *src, uint32_t *dst, size_t words, int wordsize)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= words)
        return;
    const uint32_t v = src[i];
    dst[i] = wordsize == 2 ? ((v & 0x00ff00ffu) << 8) | ((v & 0xff00ff00u) >> 8)
                           : __builtin_bswap32(v);
}

extern "C" int plh_launch_swap_words(plh_stream s, const void *src, void *dst, size_t words, int wordsize)
{
    if (!words)
        return 0;
    const size_t groups = (words + 255) / 256;
    hipLaunchKernelGGL(k_swap_words, dim3((unsigned) groups), dim3(256), 0, (hipStream_t) s,
                       (const uint32_t *) src, (uint32_t *) dst, words, wordsize);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : -(int) err;
}

// pl_frame_clear_tiles (src/renderer.c:4116-4170): the reference fills a plane through a fragment
// shader -- outcoord = gl_FragCoord.xy * (1 / size_x, 1 / size_y); tile = lessThan(fract(outcoord),
// 0.5); color.rgb = tile.x == tile.y ? c0 : c1; color.a = 1 -- here a kernel of its own: the same
// fp32 arithmetic on gl_FragCoord = texel + 1/2, one texel per lane, stores through plh_store.
__global__ void k_clear_tiles(const plh_view dst, float4_t c0, float4_t c1, float kx, float ky)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= dst.w || y >= dst.h)
        return;
    const float ox = ((float) x + 0.5f) * kx, oy = ((float) y + 0.5f) * ky;
    const bool tx = ox - __builtin_floorf(ox) < 0.5f, ty = oy - __builtin_floorf(oy) < 0.5f;
    plh_store(dst, x, y, tx == ty ? c0 : c1);
}

extern "C" int plh_launch_clear_tiles(plh_stream s, const struct plh_view *dst, const float c0[4],
                                      const float c1[4], float kx, float ky)
{
    const dim3 block(64, 4), grid((dst->w + 63) / 64, (dst->h + 3) / 4);
    const float4_t a = { c0[0], c0[1], c0[2], c0[3] }, b = { c1[0], c1[1], c1[2], c1[3] };
    hipLaunchKernelGGL(k_clear_tiles, grid, block, 0, (hipStream_t) s, *dst, a, b, kx, ky);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : -(int) err;
}
