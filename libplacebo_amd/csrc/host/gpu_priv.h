/*
 * libplacebo-hip — private object layouts behind pl_gpu / pl_tex / pl_buf /
 * pl_timer. Public struct first, private fields after (the reference uses the
 * same "public + PL_PRIV" trick, src/pl_alloc.h:84-97).
 */
#ifndef PLH_GPU_PRIV_H_
#define PLH_GPU_PRIV_H_

#include <libplacebo/gpu.h>
#include <libplacebo/hip.h>

#include "host_common.h"
#include "../hip/backend.h"

struct fmt_priv {
    struct pl_fmt_t pub;
    int plh;            // enum plh_fmt
};

// The backend half of a pl_gpu, after the reference's `struct pl_gpu_fns` (src/gpu.h:36-77). The
// public pl_tex_* / pl_buf_* / pl_pass_* / pl_timer_* entry points (gpu.c) check the API contract
// the way src/gpu.c does, complete the parameters (inferred rects, pitches) and call through
// this table; a backend function may assume what the front-end established. gpu_hip.c is the one
// implementation.
struct pl_shader_t;
struct plh_gpu_fns {
    void (*destroy)(pl_gpu gpu);
    pl_tex (*tex_create)(pl_gpu gpu, const struct pl_tex_params *params);
    void (*tex_destroy)(pl_gpu gpu, pl_tex tex);
    void (*tex_invalidate)(pl_gpu gpu, pl_tex tex);
    void (*tex_clear_ex)(pl_gpu gpu, pl_tex dst, const union pl_clear_color color);
    // rects complete, inside the textures, formats compatible
    void (*tex_blit)(pl_gpu gpu, const struct pl_tex_blit_params *params);
    // rc / row_pitch complete and checked, exactly one of ptr / buf, buffer range checked
    bool (*tex_upload)(pl_gpu gpu, const struct pl_tex_transfer_params *params);
    bool (*tex_download)(pl_gpu gpu, const struct pl_tex_transfer_params *params);
    bool (*tex_poll)(pl_gpu gpu, pl_tex tex, uint64_t timeout);
    pl_buf (*buf_create)(pl_gpu gpu, const struct pl_buf_params *params);
    void (*buf_destroy)(pl_gpu gpu, pl_buf buf);
    void (*buf_write)(pl_gpu gpu, pl_buf buf, size_t offset, const void *data, size_t size);
    bool (*buf_read)(pl_gpu gpu, pl_buf buf, size_t offset, void *dest, size_t size);
    void (*buf_copy)(pl_gpu gpu, pl_buf dst, size_t dst_offset, pl_buf src, size_t src_offset,
                     size_t size);
    bool (*buf_copy_swap)(pl_gpu gpu, pl_buf dst, size_t dst_offset, pl_buf src, size_t src_offset,
                          size_t size, int wordsize);
    bool (*buf_export)(pl_gpu gpu, pl_buf buf);
    bool (*buf_poll)(pl_gpu gpu, pl_buf buf, uint64_t timeout);
    // `recorded`: the live shader params->glsl_shader names (resolved by the front-end)
    pl_pass (*pass_create)(pl_gpu gpu, const struct pl_pass_params *params,
                           struct pl_shader_t *recorded);
    void (*pass_destroy)(pl_gpu gpu, pl_pass pass);
    // target / rc: the image the pass writes and the rect inside it (NULL: a pass without one)
    void (*pass_run)(pl_gpu gpu, const struct pl_pass_run_params *params, pl_tex target,
                     pl_rect2d rc);
    pl_timer (*timer_create)(pl_gpu gpu);
    void (*timer_destroy)(pl_gpu gpu, pl_timer timer);
    uint64_t (*timer_query)(pl_gpu gpu, pl_timer timer);
    void (*gpu_flush)(pl_gpu gpu);
    void (*gpu_finish)(pl_gpu gpu);
    bool (*gpu_is_failed)(pl_gpu gpu);
};

#define PLH_STAGE_SLOTS 8
#define PLH_STAGE_BYTES (64 * 1024)

#define PLH_FENCES 16
// compute units of the measuring stream's CU mask (0 = no mask); measured in profiles/r06_*
#define PLH_MEASURE_CUS_DEFAULT 0

struct gpu_priv {
    struct pl_gpu_t gpu;
    const struct plh_gpu_fns *fns;
    struct pl_hip_t hip;
    struct plh_dev_info info;
    int device;
    plh_stream stream;
    bool own_stream;
    // pl_hip_params.async_measure: a second stream for the per-frame measurement pass
    // (index 1 in the functions below; index 0 is `stream`). Created on first use.
    bool async_measure;
    int measure_cus;    // compute units the second stream may use (0: all; PL_HIP_MEASURE_CUS)
    plh_stream aux;
    bool aux_announced;
    // (gpu_hip.c "two streams") launches counted per stream, how far each is known to have got,
    // how far each is already ordered behind the other, and the events recorded so far
    uint64_t seq[2], done[2], after[2];
    struct plh_fence { plh_event ev; int on; uint64_t seq; bool live; } fence[PLH_FENCES];
    unsigned fence_next;
    bool failed;
    pl_cache cache;     // pl_gpu_set_cache (borrowed)
    pl_hip_peak_exchange_fn peak_exchange;  // pl_hip_set_peak_exchange
    void *peak_exchange_priv;
    struct fmt_priv fmt_store[16];
    pl_fmt fmts[16];
    // pinned staging slots for small host -> device uploads (per-frame tables: tone curve,
    // constant blocks): the copy is queued and the call returns, no stream wait
    struct {
        void *host;
        plh_event done;
        bool in_flight;
    } stage[PLH_STAGE_SLOTS];
    int stage_next;
    // Small per-pass tables that no object of the caller's outlives (Dolby Vision reshaping
    // curves: pl_shader_decode_color has no state parameter): a ring of device slots, written on
    // the main stream. A slot is reused PLH_SCRATCH_SLOTS uploads later -- behind, in stream
    // order, the pass that read it, provided that pass was dispatched by then.
    void *scratch;
    unsigned scratch_next;
    bool scratch_live[32];      // (PLH_SCRATCH_SLOTS) a recorded shader still points at the slot
};

#define PLH_SCRATCH_SLOTS 32
#define PLH_SCRATCH_BYTES 4096
// device pointer to a copy of `data` (size <= PLH_SCRATCH_BYTES); NULL on failure
const void *plh_gpu_upload_scratch(pl_gpu gpu, const void *data, size_t size);
// the shader that asked for `slot` has been dispatched, reset or freed: the slot may be reused
void plh_gpu_release_scratch(pl_gpu gpu, const void *slot);

struct tex_priv {
    struct pl_tex_t tex;
    pl_gpu gpu;
    void *ptr;
    size_t pitch;
    int plh_fmt;
    bool owned;
    // cross-stream ordering (async_measure only; all zero otherwise): launch numbers of the last
    // write (on stream `write_on`) and of the last read per stream
    uint64_t write_seq, read_seq[2];
    int write_on;
    // the renderer's measured intermediate (written on the measuring stream, read on the main one,
    // rewritten a few frames on): the launch that reads it carries a fence (plh_gpu_fence_for_launch)
    bool fence_reads;
};

struct buf_priv {
    struct pl_buf_t buf;
    void *ptr;
};

#define PLH_TIMER_RING 16
struct pl_timer_t {
    plh_event start[PLH_TIMER_RING], stop[PLH_TIMER_RING];
    unsigned head, tail;
};

#define GPU_PRIV(g)  ((struct gpu_priv *) (g))
#define TEX_PRIV(t)  ((struct tex_priv *) (t))
#define BUF_PRIV(b)  ((struct buf_priv *) (b))
#define FMT_PRIV(f)  ((const struct fmt_priv *) (f))
#define GPU_FNS(g)   (GPU_PRIV(g)->fns)

void plh_tex_view(pl_tex tex, struct plh_view *out);
// two-colour tiles over the whole texture (k_clear_tiles; kx, ky = 1 / tile period in texels)
void plh_tex_clear_tiles(pl_gpu gpu, pl_tex dst, const float c0[4], const float c1[4],
                         float kx, float ky);
void plh_timer_begin(pl_gpu gpu, pl_timer t, int on);    // `on`: stream index, see below
void plh_timer_end(pl_gpu gpu, pl_timer t, int on);

pl_cache plh_gpu_cache(pl_gpu gpu);
// unvalidated buffer IO for the library's own device-only tables (gpu.c)
void plh_buf_write(pl_gpu gpu, pl_buf buf, size_t buf_offset, const void *data, size_t size);
bool plh_buf_read(pl_gpu gpu, pl_buf buf, size_t buf_offset, void *dest, size_t size);
// runs the installed cross-GPU exchange (if any) on a finished peak measurement
void plh_gpu_peak_exchange(pl_gpu gpu, void *words, size_t size);
static inline bool plh_gpu_has_peak_exchange(pl_gpu gpu);
static inline bool plh_gpu_has_peak_exchange(pl_gpu gpu) { return !!((struct gpu_priv *) gpu)->peak_exchange; }
static inline plh_stream plh_gpu_stream(pl_gpu gpu) { return GPU_PRIV(gpu)->stream; }

// ---- two-stream ordering (gpu_hip.c). `on`: 0 = the main stream, 1 = the measurement stream.
// Everything here does nothing (and plh_gpu_stream_n returns the main stream) unless
// pl_hip_params.async_measure is set.
bool plh_gpu_async(pl_gpu gpu);
plh_stream plh_gpu_stream_n(pl_gpu gpu, int on);
// Before a launch on stream `on` that reads `reads` and writes `writes` (either may be NULL):
// makes the stream wait for whatever the other stream still has to do with them, and notes the
// launch on both textures. Returns the launch's number on its stream.
uint64_t plh_tex_order(pl_gpu gpu, int on, pl_tex reads, pl_tex writes);
// an event to ride on the launch that plh_tex_order has just counted (NULL: none wanted), and what
// became of it (gpu_hip.c)
plh_event plh_gpu_fence_for_launch(pl_gpu gpu, int on, pl_tex reads);
void plh_tex_fence_reads(pl_tex tex);  // (renderer: the measured intermediate)
void plh_gpu_fence_launched(pl_gpu gpu, int on, bool taken);
// `tex` has been read by launches on `on` that plh_tex_order did not see, all of them queued by now
void plh_tex_read_so_far(pl_gpu gpu, pl_tex tex, int on);
// a number for something just queued on `on` that is no texture (a table upload)...
uint64_t plh_gpu_stamp(pl_gpu gpu, int on);

// The reference's internal byte-swapping copy (src/gpu.h:137-165; src/gpu/utils.c:1065 implements it
// as a GLSL compute pass through the GPU's own dispatch, which this backend does not have: a kernel
// of its own here). Its tests call it (src/tests/gpu_tests.c:102-125), so the symbol is exported.
struct pl_buf_copy_swap_params {
    pl_buf src;             // must be `storable`
    size_t src_offset;
    pl_buf dst;             // must be `storable`; may be `src` (same offset: in place)
    size_t dst_offset;
    size_t size;            // bytes, a multiple of 4
    int wordsize;           // 2: swap the bytes of every 16-bit word, 4: of every 32-bit word
};
PL_API bool pl_buf_copy_swap(pl_gpu gpu, const struct pl_buf_copy_swap_params *params);
// ... which stream `on` (the other one) has to wait for
void plh_gpu_order_after(pl_gpu gpu, int on, uint64_t other_seq);
// the host has seen the result of launch `seq` of stream `on`
void plh_gpu_reached(pl_gpu gpu, int on, uint64_t seq);
void plh_gpu_sync_all(pl_gpu gpu);
static inline int plh_gpu_device(pl_gpu gpu) { return GPU_PRIV(gpu)->device; }

#endif // PLH_GPU_PRIV_H_
