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

/* ---- the PQ transfer pair as piecewise cubics (pqseg.hiph) ------------------------------------- */
#include <math.h>
#include <mutex>
#include "pqseg.hiph"

// The closed forms, in double, from the constants the ops carry (fp32, as the reference's shader
// text prints them: src/shaders/colorspace.c:643-668, 745-775 and :1792-1799, :1985-1995) -- the
// same functions pqmath.hiph evaluates in fp32: c1 = 1 - a, c2 = c3 + a.
static double pqseg_oetf(double x, double m1, double c3, double m2)
{
    const double a = (double) PQ_A;
    const double y = x > 0.0 ? pow(x, m1) : 0.0;
    const double w = a * (1.0 - y) / (1.0 + c3 * y);
    return exp(m2 * log1p(-w));
}

static double pqseg_eotf(double v, double c3, double inv_m2, double inv_m1)
{
    const double a = (double) PQ_A;
    if (!(v > 0.0))
        return 0.0;
    const double u = -expm1(log(v) * inv_m2);
    double inner = (a - u) / (a + c3 * u);
    if (!(inner > 0.0))
        return 0.0;
    return pow(inner, inv_m1);
}

// the cubic through f at four nodes of [0, 1): coefficients of 1, u, u^2, u^3 (Newton's divided
// differences, expanded)
static void pqseg_fit(const double node[4], const double f[4], float out[4])
{
    double d[4] = { f[0], f[1], f[2], f[3] };
    for (int j = 1; j < 4; j++) {
        for (int i = 3; i >= j; i--)
            d[i] = (d[i] - d[i - 1]) / (node[i] - node[i - j]);
    }
    // p(u) = d0 + (u - n0) (d1 + (u - n1) (d2 + (u - n2) d3))
    double c[4] = { d[3], 0.0, 0.0, 0.0 };      // (highest power first while multiplying out)
    int deg = 0;
    for (int j = 2; j >= 0; j--) {
        // c := c * (u - node[j]) + d[j]
        double n[4] = { 0.0, 0.0, 0.0, 0.0 };
        for (int i = 0; i <= deg; i++) {
            n[i] += c[i];
            n[i + 1] -= c[i] * node[j];
        }
        deg++;
        n[deg] += d[j];
        for (int i = 0; i <= deg; i++)
            c[i] = n[i];
    }
    out[0] = (float) c[3];
    out[1] = (float) c[2];
    out[2] = (float) c[1];
    out[3] = (float) c[0];
}

// consts = { m1, c3, m2, 1 / m2, 1 / m1 } (struct pq_consts before its log2(e)); out: PQSEG_N pieces
// of four floats -- OETF pieces first, then the EOTF's fine and coarse ones (pqseg.hiph)
static void plh_pqseg_build(const float consts[5], float *out)
{
    const double m1 = consts[0], c3 = consts[1], m2 = consts[2], inv_m2 = consts[3], inv_m1 = consts[4];
    double cheb[4];
    for (int k = 0; k < 4; k++)
        cheb[k] = 0.5 - 0.5 * cos((2 * k + 1) * M_PI / 8.0);
    for (int s = 0; s < PQSEG_O_N; s++) {
        double f[4];
        for (int k = 0; k < 4; k++)
            f[k] = pqseg_oetf(exp2((double) (PQSEG_O_T0 + s) + cheb[k]), m1, c3, m2);
        pqseg_fit(cheb, f, out + 4 * s);
    }
    float *eo = out + 4 * PQSEG_O_N;
    for (int s = 0; s < PQSEG_E_N; s++) {
        const bool lo = s < PQSEG_E_LO_N;
        const double h = lo ? (double) PQSEG_E_SPLIT / (128.0 * PQSEG_E_LO_N) : 1.0 / 128.0;
        const double v0 = lo ? s * h : (s - PQSEG_E_LO_N + PQSEG_E_SPLIT) * h;
        // (the first piece starts AT zero, where the curve is exactly zero: black stays black)
        const double first[4] = { 0.0, 0.3, 0.65, 0.95 };
        const double *node = s == 0 ? first : cheb;
        double f[4];
        for (int k = 0; k < 4; k++)
            f[k] = pqseg_eotf(v0 + h * node[k], c3, inv_m2, inv_m1);
        pqseg_fit(node, f, eo + 4 * s);
    }
}

// Test hook (tests/test_pqseg.py, CPU): the tables as the host builds them
extern "C" __attribute__((visibility("default")))
void plh_test_pqseg_build(const float consts[5], float *out)
{
    plh_pqseg_build(consts, out);
}

// The tables on the device that owns `s`, built and uploaded on first use (one per device and set
// of constants -- in practice one: the constants are SMPTE ST 2084's), kept for the life of the
// process. NULL = not available (allocation failed): the kernels then keep the closed forms.
struct pqseg_slot { int dev; float consts[5]; void *ptr; };
static std::mutex g_pqseg_mutex;
static pqseg_slot g_pqseg[16];
static int g_pqseg_n;

extern "C" const void *plh_pqseg_tables(plh_stream s, const float consts[5])
{
    const int dev = plh_stream_device(s, NULL);
    std::lock_guard<std::mutex> lock(g_pqseg_mutex);
    for (int i = 0; i < g_pqseg_n; i++) {
        if (g_pqseg[i].dev == dev && !memcmp(g_pqseg[i].consts, consts, sizeof(g_pqseg[i].consts)))
            return g_pqseg[i].ptr;
    }
    if (g_pqseg_n == 16)
        return NULL;
    static float host[PQSEG_N * 4];
    plh_pqseg_build(consts, host);
    int cur = dev;
    (void) hipGetDevice(&cur);
    if (cur != dev && hipSetDevice(dev) != hipSuccess)
        return NULL;
    void *p = NULL;
    bool ok = hipMalloc(&p, PQSEG_BYTES) == hipSuccess;
    // (a blocking copy on the null stream: once per process and device)
    ok = ok && hipMemcpy(p, host, PQSEG_BYTES, hipMemcpyHostToDevice) == hipSuccess;
    if (cur != dev)
        (void) hipSetDevice(cur);
    if (!ok) {
        if (p)
            (void) hipFree(p);
        p = NULL;
    }
    pqseg_slot &e = g_pqseg[g_pqseg_n++];
    e.dev = dev;
    memcpy(e.consts, consts, sizeof(e.consts));
    e.ptr = p;
    return p;
}

// Test hook (tests/test_gpu_pqseg.py): the pieces evaluated ON THE DEVICE, by the very functions the
// chain kernels call (pq_eotf_seg / pq_oetf_seg off a copy of the tables staged in LDS), for n values
// -- which = 0: EOTF, 1: OETF. The host test holds the result to its own emulation of the lookup
// (tests/test_pqseg.py), i.e. to what is pinned against float64 there.
__global__ __launch_bounds__(256)
void k_test_pqseg(const float *in, float *out, int n, int which, const void *tables, pq_consts k)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char pqt_smem[];
    pq_seg_stage(pqt_smem, tables, 0, threadIdx.x, 256);
    __syncthreads();
    const pq_seg seg = pq_seg_view(pqt_smem, 0, threadIdx.x & 63u, true);
    const int i = blockIdx.x * 256 + threadIdx.x;
    float v[1] = { in[min(i, n - 1)] };
    if (which == 0)
        pq_eotf_seg<1>(v, seg, k);
    else
        pq_oetf_seg<1>(v, seg, k);
    if (i < n)
        out[i] = v[0];
}

extern "C" __attribute__((visibility("default")))
int plh_test_pqseg_eval(const float consts[5], const float *in, float *out, int n, int which)
{
    const void *tables = plh_pqseg_tables(NULL, consts);
    if (!tables || n <= 0)
        return -1;
    float *din = NULL, *dout = NULL;
    if (hipMalloc((void **) &din, (size_t) n * 4) != hipSuccess || hipMalloc((void **) &dout, (size_t) n * 4) != hipSuccess)
        return -2;
    int rc = 0;
    if (hipMemcpy(din, in, (size_t) n * 4, hipMemcpyHostToDevice) != hipSuccess)
        rc = -3;
    const pq_consts k = { consts[0], consts[1], consts[2] * 1.44269504088896340736f, consts[3], consts[4] };
    if (!rc) {
        hipLaunchKernelGGL(k_test_pqseg, dim3((n + 255) / 256), dim3(256), PQSEG_BYTES, 0, din, dout, n, which, tables, k);
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out, dout, (size_t) n * 4, hipMemcpyDeviceToHost) != hipSuccess)
            rc = -4;
    }
    (void) hipFree(din);
    (void) hipFree(dout);
    return rc;
}
