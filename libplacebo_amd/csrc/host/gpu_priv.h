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

#define PLH_STAGE_SLOTS 8
#define PLH_STAGE_BYTES (64 * 1024)

struct gpu_priv {
    struct pl_gpu_t gpu;
    struct pl_hip_t hip;
    struct plh_dev_info info;
    int device;
    plh_stream stream;
    bool own_stream;
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
};

struct tex_priv {
    struct pl_tex_t tex;
    pl_gpu gpu;
    void *ptr;
    size_t pitch;
    int plh_fmt;
    bool owned;
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

void plh_tex_view(pl_tex tex, struct plh_view *out);
void plh_timer_begin(pl_gpu gpu, pl_timer t);
void plh_timer_end(pl_gpu gpu, pl_timer t);

pl_cache plh_gpu_cache(pl_gpu gpu);
// unvalidated buffer IO for the library's own device-only tables (gpu.c)
void plh_buf_write(pl_gpu gpu, pl_buf buf, size_t buf_offset, const void *data, size_t size);
bool plh_buf_read(pl_gpu gpu, pl_buf buf, size_t buf_offset, void *dest, size_t size);
// runs the installed cross-GPU exchange (if any) on a finished peak measurement
void plh_gpu_peak_exchange(pl_gpu gpu, void *words, size_t size);
static inline bool plh_gpu_has_peak_exchange(pl_gpu gpu);
static inline bool plh_gpu_has_peak_exchange(pl_gpu gpu) { return !!((struct gpu_priv *) gpu)->peak_exchange; }
static inline plh_stream plh_gpu_stream(pl_gpu gpu) { return GPU_PRIV(gpu)->stream; }
static inline int plh_gpu_device(pl_gpu gpu) { return GPU_PRIV(gpu)->device; }

#endif // PLH_GPU_PRIV_H_
