/*
 * libplacebo-hip: the HIP / CDNA4 backend constructor.
 *
 * Sits next to the reference's vulkan.h / opengl.h / d3d11.h and follows the
 * same pattern (opengl.h:33-42,114-122): a backend object that owns a pl_gpu.
 * All work is queued on one HIP stream; textures are pitched linear arrays in
 * HBM (no texture units are used — see DESIGN.md).
 */
#ifndef LIBPLACEBO_HIP_H_
#define LIBPLACEBO_HIP_H_

#include <libplacebo/gpu.h>

PL_API_BEGIN

typedef const struct pl_hip_t {
    pl_gpu gpu;
    int device;         // HIP device ordinal
    void *stream;       // hipStream_t all work is queued on
    const char *arch;   // e.g. "gfx950:sramecc+:xnack-"
    int compute_units;
} *pl_hip;

struct pl_hip_params {
    int device;         // HIP device ordinal to use
    // Optional externally owned hipStream_t (e.g. torch's current stream).
    // If NULL, a private non-blocking stream is created.
    void *stream;
    // Report limits.max_shmem_size = this many bytes (0 = 65536). Generic code
    // uses it to decide e.g. whether error diffusion fits in LDS
    // (renderer.c:2290); CDNA4 allows up to 163840.
    size_t max_shmem_size;
    // Run the HDR measurement pass of a frame (pl_shader_detect_peak: source plane -> FBO +
    // brightness statistics) on a second HIP stream, so that it overlaps the previous frame's
    // scaling / colour-mapping pass, which is still running when the next pl_render_image call
    // arrives. The counterpart of pl_vulkan_params.async_compute (vulkan.h:270-272) and, like it,
    // ON by default (PL_HIP_DEFAULTS). Results are identical; the renderer keeps its own
    // intermediates for that pass. Ignored (one stream) when `stream` is given or a peak
    // exchange is installed. The environment variable PL_HIP_ASYNC_MEASURE=0|1 overrides it.
    bool async_measure;
};

// Default values of `pl_hip_params` (the pattern of PL_VULKAN_DEFAULTS, vulkan.h:285-292): a
// later designated initialiser in pl_hip_params(...) overrides them.
#define PL_HIP_DEFAULTS     \
    .async_measure = true,

#define pl_hip_params(...) (&(struct pl_hip_params) { PL_HIP_DEFAULTS __VA_ARGS__ })
PL_API extern const struct pl_hip_params pl_hip_default_params;

// Number of HIP devices visible to this process (0 if no GPU / no runtime).
PL_API int pl_hip_device_count(void);

// Creates the backend; NULL (and a log message) on failure. Never falls back
// to a CPU path: without a usable HIP device there is no pl_gpu.
PL_API pl_hip pl_hip_create(pl_log log, const struct pl_hip_params *params);
PL_API void pl_hip_destroy(pl_hip *hip);
PL_API pl_hip pl_hip_get(pl_gpu gpu);

// Wrap an existing device allocation (e.g. a torch tensor) as a pl_tex,
// like pl_opengl_wrap / pl_vulkan_wrap. The memory is borrowed, not owned.
struct pl_hip_wrap_params {
    void *ptr;          // device pointer to texel (0,0)
    int width, height;
    size_t row_pitch;   // bytes (0 = tightly packed)
    pl_fmt format;
};

#define pl_hip_wrap_params(...) (&(struct pl_hip_wrap_params) { __VA_ARGS__ })
PL_API pl_tex pl_hip_wrap(pl_gpu gpu, const struct pl_hip_wrap_params *params);

// Device pointer / pitch of a texture created by this backend.
PL_API void *pl_hip_tex_ptr(pl_tex tex, size_t *out_row_pitch);
// Device pointer of a buffer created by this backend.
PL_API void *pl_hip_buf_ptr(pl_buf buf);

/* ---- multi-GPU: one scene rendered by several GPUs --------------------------
 * Streams are independent; the only state worth sharing is the HDR peak measurement when the
 * ranks render tiles or frames of ONE scene and must tone-map with one common peak. The
 * measurement is 816 x uint32 (the reference's `peak_buf_data`, shaders/colorspace.c:936-942):
 * every word is a SUM across ranks, except words [36, 48) (frame_max_pq), which are a MAX.
 * All integer, hence order-independent: every rank derives bit-identical tone curves.
 */

// Called on the host right before a finished measurement is read back, with the device
// buffer and the stream it was produced on. The callback must leave the reduced words in
// place, ordered on `stream` (or synchronise itself).
typedef void (*pl_hip_peak_exchange_fn)(void *priv, void *words, size_t size, void *stream);

// Install (fn != NULL) or remove the exchange for every measurement made on `gpu`.
PL_API void pl_hip_set_peak_exchange(pl_gpu gpu, pl_hip_peak_exchange_fn fn, void *priv);

// Ready-made exchange over RCCL (xGMI inside a node). `nccl_comm` is an initialised
// ncclComm_t whose local device is the one `gpu` runs on. `nccl_all_reduce` is the address of
// ncclAllReduce in the RCCL instance that created the communicator (a communicator is only
// valid inside the instance that made it, and that instance must sit on the same HIP runtime as
// this library: ROCm's librccl, not a copy bundled with another framework); NULL = resolve
// "ncclAllReduce" at run time (process scope, then dlopen of librccl.so). The library has no
// link-time dependency on RCCL. Typical use:
//     pl_hip_rccl x = pl_hip_rccl_create(gpu, comm, (void *) ncclAllReduce);
//     pl_hip_set_peak_exchange(gpu, pl_hip_rccl_peak_exchange, x);
typedef struct pl_hip_rccl_t *pl_hip_rccl;
PL_API pl_hip_rccl pl_hip_rccl_create(pl_gpu gpu, void *nccl_comm, void *nccl_all_reduce);
PL_API void pl_hip_rccl_destroy(pl_hip_rccl *x);
PL_API void pl_hip_rccl_peak_exchange(void *priv, void *words, size_t size, void *stream);
// number of exchanges performed / nonzero RCCL status seen so far
PL_API int pl_hip_rccl_stats(pl_hip_rccl x, int *out_errors);

PL_API_END

#endif // LIBPLACEBO_HIP_H_
