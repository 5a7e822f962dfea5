/*
 * libplacebo-hip -- the HIP implementation of the pl_gpu backend table (gpu_priv.h: struct
 * plh_gpu_fns; the reference's counterpart of this file is a backend such as src/dummy.c or
 * src/opengl/gpu*.c). Object creation, the format table, limits, and the functions the validating
 * front-end (gpu.c) calls through. Textures are pitched linear device arrays; all work is ordered
 * on one HIP stream, so uploads / passes / downloads issued in API order execute in that order
 * without further fences.
 */
#include <stdlib.h>
#include <string.h>

#include <libplacebo/hip.h>

#include "gpu_priv.h"
#include "shaders_priv.h"
#include "cache_priv.h"

const struct pl_hip_params pl_hip_default_params = { PL_HIP_DEFAULTS };
static const struct plh_gpu_fns hip_fns;

/* ------------------------------------------------------------------------ */
/* format table                                                              */

#define CAPS_ALL (PL_FMT_CAP_SAMPLEABLE | PL_FMT_CAP_STORABLE | PL_FMT_CAP_LINEAR |    \
                  PL_FMT_CAP_RENDERABLE | PL_FMT_CAP_BLENDABLE | PL_FMT_CAP_BLITTABLE | \
                  PL_FMT_CAP_HOST_READABLE | PL_FMT_CAP_READWRITE)

#define FMT(nm, ty, n, bits, plhfmt, vtx, gtype, gfmt) {                                   \
    .pub = {                                                                            \
        .name = nm, .type = ty, .num_components = n,                                    \
        .caps = CAPS_ALL | ((vtx) ? PL_FMT_CAP_VERTEX : 0),                             \
        .component_depth = { bits, (n) > 1 ? bits : 0, (n) > 2 ? bits : 0, (n) > 3 ? bits : 0 }, \
        .host_bits       = { bits, (n) > 1 ? bits : 0, (n) > 2 ? bits : 0, (n) > 3 ? bits : 0 }, \
        .sample_order = {0, 1, 2, 3},                                                   \
        .internal_size = (n) * (bits) / 8, .texel_size = (n) * (bits) / 8,              \
        .texel_align = (bits) / 8, .gatherable = true,                                  \
        .glsl_type = gtype, .glsl_format = gfmt,                                        \
    }, .plh = plhfmt }

static const struct fmt_priv fmt_table[] = {
    FMT("r8",       PL_FMT_UNORM, 1,  8, PLH_FMT_R8,      false, "float", "r8"),
    FMT("rg8",      PL_FMT_UNORM, 2,  8, PLH_FMT_RG8,     false, "vec2",  "rg8"),
    FMT("rgba8",    PL_FMT_UNORM, 4,  8, PLH_FMT_RGBA8,   false, "vec4",  "rgba8"),
    FMT("r16",      PL_FMT_UNORM, 1, 16, PLH_FMT_R16,     false, "float", "r16"),
    FMT("rg16",     PL_FMT_UNORM, 2, 16, PLH_FMT_RG16,    false, "vec2",  "rg16"),
    FMT("rgba16",   PL_FMT_UNORM, 4, 16, PLH_FMT_RGBA16,  false, "vec4",  "rgba16"),
    FMT("r16hf",    PL_FMT_FLOAT, 1, 16, PLH_FMT_R16F,    false, "float", "r16f"),
    FMT("rg16hf",   PL_FMT_FLOAT, 2, 16, PLH_FMT_RG16F,   false, "vec2",  "rg16f"),
    FMT("rgba16hf", PL_FMT_FLOAT, 4, 16, PLH_FMT_RGBA16F, false, "vec4",  "rgba16f"),
    FMT("r32f",     PL_FMT_FLOAT, 1, 32, PLH_FMT_R32F,    true,  "float", "r32f"),
    FMT("rg32f",    PL_FMT_FLOAT, 2, 32, PLH_FMT_RG32F,   true,  "vec2",  "rg32f"),
    FMT("rgba32f",  PL_FMT_FLOAT, 4, 32, PLH_FMT_RGBA32F, true,  "vec4",  "rgba32f"),
};

#define NUM_FMTS ((int) PL_ARRAY_SIZE(fmt_table))

// Same ordering rule as the reference's pl_gpu_finalize (gpu/utils.c:26-81);
// all our formats share caps, so this reduces to "lower depth first, then name"
static int cmp_fmt(const void *pa, const void *pb)
{
    pl_fmt a = *(pl_fmt *) pa, b = *(pl_fmt *) pb;
    for (int i = 0; i < 4; i++) {
        if (a->component_depth[i] != b->component_depth[i])
            return a->component_depth[i] < b->component_depth[i] ? -1 : 1;
        if (a->host_bits[i] != b->host_bits[i])
            return a->host_bits[i] < b->host_bits[i] ? -1 : 1;
    }
    return strcmp(a->name, b->name);
}


/* Test hook (tests/test_render_plan.py): a format description without a device, so that the
 * renderer's planner can be exercised on CPU-only hosts. */
PL_API pl_fmt plh_test_format(const char *name);
pl_fmt plh_test_format(const char *name)
{
    for (int i = 0; i < NUM_FMTS; i++) {
        if (!strcmp(fmt_table[i].pub.name, name))
            return &fmt_table[i].pub;
    }
    return NULL;
}

/* ------------------------------------------------------------------------ */
/* backend object                                                            */

int pl_hip_device_count(void)
{
    return plh_dev_count();
}

pl_hip pl_hip_create(pl_log log, const struct pl_hip_params *params)
{
    params = PL_DEF(params, &pl_hip_default_params);
    if (plh_dev_count() <= params->device) {
        pl_msg(log, PL_LOG_FATAL, "pl_hip_create: no HIP device %d (found %d). "
               "There is no CPU fallback for this backend.",
               params->device, plh_dev_count());
        return NULL;
    }

    struct gpu_priv *p = calloc(1, sizeof(*p));
    if (!p)
        return NULL;

    int err = plh_dev_open(params->device, &p->info);
    if (err) {
        pl_msg(log, PL_LOG_FATAL, "pl_hip_create: opening device %d failed: %s",
               params->device, plh_strerror(err));
        free(p);
        return NULL;
    }

    p->device = params->device;
    if (params->stream) {
        p->stream = params->stream;
    } else {
        err = plh_stream_create(p->device, &p->stream);
        if (err) {
            pl_msg(log, PL_LOG_FATAL, "pl_hip_create: stream creation failed: %s",
                   plh_strerror(err));
            free(p);
            return NULL;
        }
        p->own_stream = true;
    }

    p->fns = &hip_fns;
    p->async_measure = params->async_measure && !params->stream;   // (own streams only)
    // PL_HIP_ASYNC_MEASURE=0|1 overrides the parameter: lets a whole test suite or an unmodified
    // application run with the option on
    const char *async_env = getenv("PL_HIP_ASYNC_MEASURE");
    if (async_env)
        p->async_measure = atoi(async_env) && !params->stream;
    // PL_HIP_MEASURE_CUS=n: the second stream is created with a CU mask of n units (n / 8 per XCD)
    const char *cus_env = getenv("PL_HIP_MEASURE_CUS");
    p->measure_cus = cus_env ? atoi(cus_env) : PLH_MEASURE_CUS_DEFAULT;
    struct pl_gpu_t *gpu = &p->gpu;
    gpu->log = log;
    gpu->glsl = (struct pl_glsl_version) {
        .version = 450,
        .vulkan = true,
        .compute = true,
        .max_shmem_size = PL_DEF(params->max_shmem_size, 160 * 1024), // CDNA4 LDS per CU
        .max_group_threads = 1024,
        .max_group_size = { 1024, 1024, 1024 },
        .subgroup_size = 64,
        .min_gather_offset = -32,
        .max_gather_offset = 31,
    };
    gpu->limits = (struct pl_gpu_limits) {
        .thread_safe = false,
        .callbacks = false,
        .max_buf_size = p->info.total_mem,
        .max_ubo_size = 65536,
        .max_ssbo_size = p->info.total_mem,
        .max_tex_1d_dim = 1 << 16,
        .max_tex_2d_dim = 1 << 16,
        .max_tex_3d_dim = 0,
        .buf_transfer = true,
        .align_tex_xfer_pitch = 256,
        .align_tex_xfer_offset = 256,
        .max_variable_comps = 0,
        .max_constants = 0,
        .array_size_constants = true,
        .max_pushc_size = 4096, // kernel arguments
        .max_dispatch = { 1u << 31, 65535, 65535 },
        .fragment_queues = 0,   // every pass is a compute pass (dispatch.c:1236)
        .compute_queues = 1,
    };
    memcpy(gpu->uuid, p->info.uuid, 16);
    gpu->pci = (struct pl_gpu_pci_address) {
        .domain = p->info.pci_domain, .bus = p->info.pci_bus, .device = p->info.pci_device,
    };

    for (int i = 0; i < NUM_FMTS; i++) {
        p->fmt_store[i] = fmt_table[i];
        p->fmt_store[i].pub.num_planes = 1;
        p->fmt_store[i].pub.planes[0].format = &p->fmt_store[i].pub;
        p->fmt_store[i].pub.signature = plh_mem_hash(fmt_table[i].pub.name,
                                                     strlen(fmt_table[i].pub.name));
        p->fmts[i] = &p->fmt_store[i].pub;
    }
    qsort(p->fmts, NUM_FMTS, sizeof(p->fmts[0]), cmp_fmt);
    gpu->formats = p->fmts;
    gpu->num_formats = NUM_FMTS;

    p->hip = (struct pl_hip_t) {
        .gpu = gpu,
        .device = p->device,
        .stream = p->stream,
        .arch = p->info.arch,
        .compute_units = p->info.compute_units,
    };

    pl_msg(log, PL_LOG_INFO, "pl_hip: device %d '%s' (%s), %d CUs, %zu MiB",
           p->device, p->info.name, p->info.arch, p->info.compute_units,
           p->info.total_mem >> 20);
    return &p->hip;
}

static void hip_destroy(pl_gpu gpu)
{
    struct gpu_priv *p = GPU_PRIV(gpu);
    plh_gpu_sync_all(gpu);
    if (p->aux)
        plh_stream_destroy(p->aux);
    for (int i = 0; i < PLH_FENCES; i++)
        plh_event_destroy(p->fence[i].ev);
    for (int i = 0; i < PLH_STAGE_SLOTS; i++) {
        plh_event_destroy(p->stage[i].done);
        plh_host_free(p->stage[i].host);
    }
    plh_free(p->scratch);
    if (p->own_stream)
        plh_stream_destroy(p->stream);
    free(p);
}

void pl_hip_destroy(pl_hip *hip)
{
    if (!hip || !*hip)
        return;
    pl_gpu gpu = (*hip)->gpu;
    GPU_FNS(gpu)->destroy(gpu);
    *hip = NULL;
}

pl_hip pl_hip_get(pl_gpu gpu)
{
    return gpu ? &GPU_PRIV(gpu)->hip : NULL;
}

static void hip_gpu_flush(pl_gpu gpu)
{
    (void) gpu; // HIP submits eagerly
}

static int sync_main(struct gpu_priv *p);

static void hip_gpu_finish(pl_gpu gpu)
{
    struct gpu_priv *p = GPU_PRIV(gpu);
    int err = sync_main(p);
    if (!err && p->aux && !(err = plh_stream_sync(p->aux)))
        p->done[1] = p->seq[1];
    if (err) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_gpu_finish: %s", plh_strerror(err));
        p->failed = true;
    }
}

/* ------------------------------------------------------------------------ */
/* two streams: who has to wait for whom                                      */
//
// With pl_hip_params.async_measure the measuring pass of a frame runs on a second stream while
// the main stream is still busy with the previous frame. Ordering between the two is kept with
// as few stream events as possible -- every event recorded behind a kernel costs the queue a few
// microseconds, which is what the second stream is there to save. Each stream counts the
// launches that touch a texture (`seq`); a texture remembers the count of its last write and of
// its last read per stream; the host remembers how far each stream is known to have got (`done`:
// a stream sync, a measurement whose result has been seen on the host) and how far each stream
// is already ordered behind the other (`after`). Only when a launch depends on a count that is
// neither known done nor already waited for is an event recorded (on the other stream, at its
// current end -- later than needed, never earlier) and waited on.

bool plh_gpu_async(pl_gpu gpu)
{
    return GPU_PRIV(gpu)->async_measure;
}

plh_stream plh_gpu_stream_n(pl_gpu gpu, int on)
{
    struct gpu_priv *p = GPU_PRIV(gpu);
    if (!on || !p->async_measure)
        return p->stream;
    // (a greatest- or least-priority measuring stream changes nothing: profiles/r05_04_peak_prio.txt)
    if (!p->aux && plh_stream_create_masked(p->device, p->measure_cus, &p->aux)) {
        pl_msg(gpu->log, PL_LOG_WARN, "pl_hip: no second stream: async_measure disabled");
        p->async_measure = false;
        return p->stream;
    }
    if (!p->aux_announced) {
        pl_msg(gpu->log, PL_LOG_INFO, "pl_hip: measurement passes run on a second stream (async_measure)");
        p->aux_announced = true;
    }
    return p->aux;
}

// An event for the end of the launch that is about to be issued on stream `on` (its number is
// already counted: plh_tex_order): offered to the launcher so that it rides on the dispatch itself
// (backend.h: plh_launch_offer_stop). plh_gpu_fence_launched() afterwards says whether it did; if
// so the ring entry is a fence like any other, and fence_here() at that same point of the stream
// returns it instead of queueing an event record of its own.
// Only the launches whose end somebody is going to ask for get one -- those that read a measured
// intermediate: a stop event on EVERY launch costs the short multi-pass frames more than the
// queued records cost the others (default preset 1080p -> 4K: 0.083 -> 0.091 ms, measured).
void plh_tex_fence_reads(pl_tex tex)
{
    TEX_PRIV(tex)->fence_reads = true;
}

plh_event plh_gpu_fence_for_launch(pl_gpu gpu, int on, pl_tex reads)
{
    struct gpu_priv *p = GPU_PRIV(gpu);
    if (!p->async_measure || !reads || !TEX_PRIV(reads)->fence_reads)
        return NULL;
    struct plh_fence *f = &p->fence[p->fence_next % PLH_FENCES];
    f->live = false;
    if (!f->ev && plh_event_create(&f->ev))
        return NULL;
    return f->ev;
}

void plh_gpu_fence_launched(pl_gpu gpu, int on, bool taken)
{
    struct gpu_priv *p = GPU_PRIV(gpu);
    if (!taken)
        return;     // (the slot stays free: nothing was recorded into its event)
    struct plh_fence *f = &p->fence[p->fence_next++ % PLH_FENCES];
    f->on = on;
    f->seq = p->seq[on];
    f->live = true;
}

// mark the current end of stream `on` with an event
static struct plh_fence *fence_here(pl_gpu gpu, int on)
{
    struct gpu_priv *p = GPU_PRIV(gpu);
    // (the launch that is the current end of the stream may carry its own)
    for (int i = 0; i < PLH_FENCES; i++) {
        struct plh_fence *c = &p->fence[i];
        if (c->live && c->on == on && c->seq == p->seq[on])
            return c;
    }
    // (an event may be re-recorded while a wait on its earlier recording is still queued: the
    // wait keeps the recording it was issued against)
    struct plh_fence *f = &p->fence[p->fence_next++ % PLH_FENCES];
    f->live = false;
    if (!f->ev && plh_event_create(&f->ev))
        return NULL;
    if (plh_event_record(f->ev, plh_gpu_stream_n(gpu, on)))
        return NULL;
    f->on = on;
    f->seq = p->seq[on];
    f->live = true;
    return f;
}

// stream `on` continues only once the other stream has got to its launch number `s`
static void order_after(pl_gpu gpu, int on, uint64_t s)
{
    struct gpu_priv *p = GPU_PRIV(gpu);
    const int other = !on;
    if (!s || s <= p->done[other] || s <= p->after[on])
        return;
    struct plh_fence *f = NULL;
    for (int i = 0; i < PLH_FENCES; i++) {
        struct plh_fence *c = &p->fence[i];
        if (c->live && c->on == other && c->seq >= s && (!f || c->seq < f->seq))
            f = c;
    }
    if (f && plh_event_query(f->ev) == 1) {
        p->done[other] = PL_MAX(p->done[other], f->seq);
        return;
    }
    if (!f && !(f = fence_here(gpu, other))) {
        plh_stream_sync(plh_gpu_stream_n(gpu, other));  // no event to be had: the blunt way
        p->done[other] = p->seq[other];
        return;
    }
    plh_stream_wait_event(plh_gpu_stream_n(gpu, on), f->ev);
    p->after[on] = PL_MAX(p->after[on], f->seq);
}

uint64_t plh_tex_order(pl_gpu gpu, int on, pl_tex reads, pl_tex writes)
{
    struct gpu_priv *p = GPU_PRIV(gpu);
    if (!p->async_measure)
        return 0;
    struct tex_priv *r = reads ? TEX_PRIV(reads) : NULL, *w = writes ? TEX_PRIV(writes) : NULL;
    if (r && r->write_on != on)
        order_after(gpu, on, r->write_seq);
    if (w) {
        if (w->write_on != on)
            order_after(gpu, on, w->write_seq);
        order_after(gpu, on, w->read_seq[!on]);
    }
    const uint64_t s = ++p->seq[on];
    if (r)
        r->read_seq[on] = s;
    if (w) {
        w->write_seq = s;
        w->write_on = on;
    }
    return s;
}

void plh_tex_read_so_far(pl_gpu gpu, pl_tex tex, int on)
{
    struct gpu_priv *p = GPU_PRIV(gpu);
    if (!p->async_measure || !tex)
        return;
    TEX_PRIV(tex)->read_seq[on] = p->seq[on];
    fence_here(gpu, on);
}

uint64_t plh_gpu_stamp(pl_gpu gpu, int on)
{
    struct gpu_priv *p = GPU_PRIV(gpu);
    return p->async_measure ? ++p->seq[on] : 0;
}

// Work queued on the main stream that no texture's read / write count records (buffer copies,
// staging uploads, pl_pass_run's compute path, pl_dispatch_compute) still takes a number: a fence
// that rode on an earlier launch (plh_gpu_fence_for_launch) must never pass for "the current end
// of the stream" in fence_here() once anything at all has been queued behind that launch.
static inline void queued_unnumbered(struct gpu_priv *g)
{
    if (g->async_measure)
        g->seq[0]++;
}

void plh_gpu_order_after(pl_gpu gpu, int on, uint64_t other_seq)
{
    if (GPU_PRIV(gpu)->async_measure)
        order_after(gpu, on, other_seq);
}

void plh_gpu_reached(pl_gpu gpu, int on, uint64_t seq)
{
    struct gpu_priv *p = GPU_PRIV(gpu);
    p->done[on] = PL_MAX(p->done[on], seq);
}

static int sync_main(struct gpu_priv *p)
{
    const int err = plh_stream_sync(p->stream);
    if (!err)
        p->done[0] = p->seq[0];
    return err;
}

void plh_gpu_sync_all(pl_gpu gpu)
{
    struct gpu_priv *p = GPU_PRIV(gpu);
    sync_main(p);
    if (p->aux && !plh_stream_sync(p->aux))
        p->done[1] = p->seq[1];
}

static bool hip_gpu_is_failed(pl_gpu gpu)
{
    return GPU_PRIV(gpu)->failed;
}

/* ------------------------------------------------------------------------ */
/* textures                                                                  */

static pl_tex hip_tex_create(pl_gpu gpu, const struct pl_tex_params *params)
{
    struct gpu_priv *g = GPU_PRIV(gpu);
    struct tex_priv *t = calloc(1, sizeof(*t));
    if (!t)
        return NULL;
    t->tex.params = *params;
    t->tex.params.initial_data = NULL;
    t->tex.sampler_type = PL_SAMPLER_NORMAL;
    t->gpu = gpu;
    t->plh_fmt = FMT_PRIV(params->format)->plh;
    const int rows = PL_MAX(params->h, 1);
    const size_t row_bytes = (size_t) params->w * params->format->texel_size;
    t->pitch = PL_ALIGN2(row_bytes, (size_t) 256);
    t->ptr = plh_malloc(g->device, t->pitch * rows);
    t->owned = true;
    if (!t->ptr) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_create: out of device memory (%zu bytes)",
               t->pitch * rows);
        free(t);
        return NULL;
    }

    if (params->initial_data) {
        const int err = plh_copy2d_h2d(g->stream, t->ptr, t->pitch, params->initial_data,
                                       row_bytes, row_bytes, rows);
        queued_unnumbered(g);
        // initial_data may be freed by the caller right away
        if (err || sync_main(g)) {
            pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_create: initial upload failed");
            plh_free(t->ptr);
            free(t);
            return NULL;
        }
    }
    return &t->tex;
}

pl_tex pl_hip_wrap(pl_gpu gpu, const struct pl_hip_wrap_params *params)
{
    if (!params || !params->ptr || !params->format || params->width <= 0 || params->height <= 0) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_hip_wrap: invalid parameters");
        return NULL;
    }
    struct tex_priv *t = calloc(1, sizeof(*t));
    if (!t)
        return NULL;
    t->tex.params = (struct pl_tex_params) {
        .w = params->width, .h = params->height, .format = params->format,
        .sampleable = true, .renderable = true, .storable = true,
        .blit_src = true, .blit_dst = true, .host_writable = true, .host_readable = true,
    };
    t->gpu = gpu;
    t->plh_fmt = FMT_PRIV(params->format)->plh;
    t->ptr = params->ptr;
    t->pitch = PL_DEF(params->row_pitch, (size_t) params->width * params->format->texel_size);
    t->owned = false;
    return &t->tex;
}

static void hip_tex_destroy(pl_gpu gpu, pl_tex tex)
{
    struct tex_priv *t = TEX_PRIV(tex);
    if (t->owned) {
        // the allocation may still be referenced by queued work
        plh_gpu_sync_all(gpu);
        plh_free(t->ptr);
    }
    free(t);
}

static void hip_tex_invalidate(pl_gpu gpu, pl_tex tex)
{
    (void) gpu; (void) tex; // contents become undefined: nothing to do
}

void plh_tex_view(pl_tex tex, struct plh_view *out)
{
    const struct tex_priv *t = TEX_PRIV(tex);
    *out = (struct plh_view) {
        .ptr = t->ptr, .w = tex->params.w, .h = PL_MAX(tex->params.h, 1),
        .pitch = (int32_t) t->pitch, .fmt = t->plh_fmt,
    };
}

void *pl_hip_tex_ptr(pl_tex tex, size_t *out_row_pitch)
{
    const struct tex_priv *t = TEX_PRIV(tex);
    if (out_row_pitch)
        *out_row_pitch = t->pitch;
    return t->ptr;
}

static void hip_tex_clear_ex(pl_gpu gpu, pl_tex dst, const union pl_clear_color color)
{
    struct plh_view v;
    plh_tex_view(dst, &v);
    plh_tex_order(gpu, 0, NULL, dst);
    const int err = plh_launch_clear(GPU_PRIV(gpu)->stream, &v, color.f);
    if (err)
        pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_clear: %s", plh_strerror(err));
}

// (not part of struct pl_gpu_fns: the reference has no such entry point either, it dispatches a
// shader; renderer.c's pl_frame_clear_tiles is the one caller)
void plh_tex_clear_tiles(pl_gpu gpu, pl_tex dst, const float c0[4], const float c1[4],
                         float kx, float ky)
{
    struct plh_view v;
    plh_tex_view(dst, &v);
    plh_tex_order(gpu, 0, NULL, dst);
    const int err = plh_launch_clear_tiles(GPU_PRIV(gpu)->stream, &v, c0, c1, kx, ky);
    if (err)
        pl_msg(gpu->log, PL_LOG_ERR, "pl_frame_clear_tiles: %s", plh_strerror(err));
}

// A blit is a pass with a bare nearest / bilinear sampler and no colour stages: the same
// kernels the renderer uses (the reference emulates blits with a compute shader the same way
// on backends without a native one, src/gpu/utils.c:852).
static void hip_tex_blit(pl_gpu gpu, const struct pl_tex_blit_params *params)
{
    pl_tex src = params->src, dst = params->dst;
    const pl_rect3d sr = params->src_rc, dr = params->dst_rc;
    const int w = abs(dr.x1 - dr.x0), h = abs(dr.y1 - dr.y0);
    struct plh_pass *pass = calloc(1, sizeof(*pass));
    if (!pass)
        return;
    struct plh_sampler_args *s = &pass->s;
    s->type = params->sample_mode == PL_TEX_SAMPLE_LINEAR ? PLH_SAMPLE_BILINEAR : PLH_SAMPLE_NEAREST;
    plh_tex_view(src, &s->src);
    const float sx = 1.0f / s->src.w, sy = 1.0f / s->src.h;
    const float x0 = sx * sr.x0, x1 = sx * sr.x1, y0 = sy * sr.y0, y1 = sy * sr.y1;
    s->pos[0][0] = x0; s->pos[0][1] = y0;
    s->pos[1][0] = x1; s->pos[1][1] = y0;
    s->pos[2][0] = x0; s->pos[2][1] = y1;
    s->pos[3][0] = x1; s->pos[3][1] = y1;
    s->pt[0] = sx;
    s->pt[1] = sy;
    s->scale = 1.0f;
    s->comp_mask = 0xf;
    s->linear = s->type == PLH_SAMPLE_BILINEAR;
    s->rect_w = abs(sr.x1 - sr.x0);
    s->rect_h = abs(sr.y1 - sr.y0);
    s->rect_on_grid = 1;

    plh_tex_view(dst, &pass->dst);
    pass->width = w;
    pass->height = h;
    pass->out_scale[0] = 1.0 / w;
    pass->out_scale[1] = 1.0 / h;
    pass->base_x = dr.x0 - (dr.x0 > dr.x1);
    pass->base_y = dr.y0 - (dr.y0 > dr.y1);
    pass->dir_x = dr.x0 > dr.x1 ? -1 : 1;
    pass->dir_y = dr.y0 > dr.y1 ? -1 : 1;
    // a 1:1 copy returns the texels themselves (what a texture unit does on the grid)
    if (s->type == PLH_SAMPLE_BILINEAR && s->rect_w == w && s->rect_h == h)
        s->type = PLH_SAMPLE_NEAREST;

    plh_tex_order(gpu, 0, src, dst);
    const int err = plh_launch_pass(GPU_PRIV(gpu)->stream, pass);
    free(pass);
    if (err) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_blit: %s", plh_strerror(err));
        GPU_PRIV(gpu)->failed = true;
    }
}

// 2-D copy between a texture region and host memory or a buffer, on the stream
static bool hip_tex_transfer(pl_gpu gpu, const struct pl_tex_transfer_params *params, bool upload)
{
    struct gpu_priv *g = GPU_PRIV(gpu);
    pl_tex tex = params->tex;
    const struct tex_priv *t = TEX_PRIV(tex);
    const pl_rect3d rc = params->rc;
    const size_t tsz = tex->params.format->texel_size;
    const size_t row_bytes = (size_t) (rc.x1 - rc.x0) * tsz;
    const size_t rows = rc.y1 - rc.y0;
    const size_t host_pitch = params->row_pitch;
    uint8_t *dev = (uint8_t *) t->ptr + (size_t) rc.y0 * t->pitch + (size_t) rc.x0 * tsz;

    plh_tex_order(gpu, 0, upload ? NULL : tex, upload ? tex : NULL);
    if (params->timer)
        plh_timer_begin(gpu, params->timer, 0);

    int err;
    if (params->buf) {
        uint8_t *bptr = (uint8_t *) BUF_PRIV(params->buf)->ptr + params->buf_offset;
        err = upload ? plh_copy2d_d2d(g->stream, dev, t->pitch, bptr, host_pitch, row_bytes, rows)
                     : plh_copy2d_d2d(g->stream, bptr, host_pitch, dev, t->pitch, row_bytes, rows);
    } else {
        err = upload ? plh_copy2d_h2d(g->stream, dev, t->pitch, params->ptr, host_pitch, row_bytes, rows)
                     : plh_copy2d_d2h(g->stream, params->ptr, host_pitch, dev, t->pitch, row_bytes, rows);
        // pageable host memory: the reference's contract is that `ptr` may be
        // reused / is filled when the call returns (gpu.h, no callback given)
        if (!err && !params->callback)
            err = sync_main(g);
    }

    if (params->timer)
        plh_timer_end(gpu, params->timer, 0);

    if (err) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_%s: %s", upload ? "upload" : "download",
               plh_strerror(err));
        g->failed = true;
        return false;
    }
    if (params->callback) {
        sync_main(g);
        params->callback(params->priv);
    }
    return true;
}

static bool hip_tex_upload(pl_gpu gpu, const struct pl_tex_transfer_params *params)
{
    return hip_tex_transfer(gpu, params, true);
}

static bool hip_tex_download(pl_gpu gpu, const struct pl_tex_transfer_params *params)
{
    return hip_tex_transfer(gpu, params, false);
}

static bool stream_busy(pl_gpu gpu, uint64_t timeout)
{
    if (timeout) {
        pl_gpu_finish(gpu);
        return false;
    }
    const struct gpu_priv *p = GPU_PRIV(gpu);
    return plh_stream_idle(p->stream) == 0 || (p->aux && plh_stream_idle(p->aux) == 0);
}

static bool hip_tex_poll(pl_gpu gpu, pl_tex tex, uint64_t timeout)
{
    (void) tex;
    return stream_busy(gpu, timeout);
}

/* ------------------------------------------------------------------------ */
/* buffers                                                                   */

static pl_buf hip_buf_create(pl_gpu gpu, const struct pl_buf_params *params)
{
    struct gpu_priv *g = GPU_PRIV(gpu);
    struct buf_priv *b = calloc(1, sizeof(*b));
    if (!b)
        return NULL;
    b->buf.params = *params;
    b->buf.params.initial_data = NULL;
    b->ptr = plh_malloc(g->device, params->size);
    if (!b->ptr) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_buf_create: out of device memory");
        free(b);
        return NULL;
    }
    if (params->initial_data) {
        plh_copy2d_h2d(g->stream, b->ptr, params->size, params->initial_data, params->size,
                       params->size, 1);
        queued_unnumbered(g);
        sync_main(g);
    }
    return &b->buf;
}

static void hip_buf_destroy(pl_gpu gpu, pl_buf buf)
{
    plh_gpu_sync_all(gpu);
    plh_free(BUF_PRIV(buf)->ptr);
    free(BUF_PRIV(buf));
}

void *pl_hip_buf_ptr(pl_buf buf)
{
    return BUF_PRIV(buf)->ptr;
}

// (also the library's own way to fill its device-only tables, without the API checks)
void plh_buf_write(pl_gpu gpu, pl_buf buf, size_t buf_offset, const void *data, size_t size)
{
    struct gpu_priv *g = GPU_PRIV(gpu);
    uint8_t *dst = (uint8_t *) BUF_PRIV(buf)->ptr + buf_offset;
    if (size <= PLH_STAGE_BYTES) {
        // through a pinned slot: `data` is the caller's again as soon as it is copied there,
        // the device copy is ordered on the stream like everything else
        const int i = g->stage_next;
        if (!g->stage[i].host) {
            g->stage[i].host = plh_host_alloc(PLH_STAGE_BYTES);
            if (g->stage[i].host && plh_event_create(&g->stage[i].done)) {
                plh_host_free(g->stage[i].host);
                g->stage[i].host = NULL;
            }
        }
        if (g->stage[i].host) {
            if (g->stage[i].in_flight)
                plh_event_sync(g->stage[i].done);   // eight uploads ago: long finished
            memcpy(g->stage[i].host, data, size);
            queued_unnumbered(g);
            if (!plh_copy2d_h2d(g->stream, dst, size, g->stage[i].host, size, size, 1) &&
                !plh_event_record(g->stage[i].done, g->stream)) {
                g->stage[i].in_flight = true;
                g->stage_next = (i + 1) % PLH_STAGE_SLOTS;
                return;
            }
        }
    }
    queued_unnumbered(g);
    plh_copy2d_h2d(g->stream, dst, size, data, size, size, 1);
    sync_main(g);
}

const void *plh_gpu_upload_scratch(pl_gpu gpu, const void *data, size_t size)
{
    struct gpu_priv *g = GPU_PRIV(gpu);
    if (size > PLH_SCRATCH_BYTES)
        return NULL;
    if (!g->scratch) {
        g->scratch = plh_malloc(g->device, (size_t) PLH_SCRATCH_SLOTS * PLH_SCRATCH_BYTES);
        if (!g->scratch)
            return NULL;
    }
    // The slots are a ring; a slot is taken until the shader that asked for it has been dispatched,
    // reset or freed (plh_gpu_release_scratch). With every slot taken -- more than PLH_SCRATCH_SLOTS
    // such shaders recorded and none of them run yet -- the upload FAILS (the shader then fails) rather
    // than hand a live slot to a second shader, whose curves the first one would silently read
    // (VERDICT r05 item 0).
    _Static_assert(PLH_SCRATCH_SLOTS <= 32, "scratch_live");
    const unsigned idx = g->scratch_next % PLH_SCRATCH_SLOTS;
    if (g->scratch_live[idx])
        return NULL;
    g->scratch_next++;
    uint8_t *slot = (uint8_t *) g->scratch + (size_t) idx * PLH_SCRATCH_BYTES;
    // through the pinned staging ring, like plh_buf_write
    const int i = g->stage_next;
    if (!g->stage[i].host) {
        g->stage[i].host = plh_host_alloc(PLH_STAGE_BYTES);
        if (g->stage[i].host && plh_event_create(&g->stage[i].done)) {
            plh_host_free(g->stage[i].host);
            g->stage[i].host = NULL;
        }
    }
    if (!g->stage[i].host)
        return NULL;
    if (g->stage[i].in_flight)
        plh_event_sync(g->stage[i].done);
    memcpy(g->stage[i].host, data, size);
    queued_unnumbered(g);
    if (plh_copy2d_h2d(g->stream, slot, size, g->stage[i].host, size, size, 1) ||
        plh_event_record(g->stage[i].done, g->stream))
        return NULL;
    g->stage[i].in_flight = true;
    g->stage_next = (i + 1) % PLH_STAGE_SLOTS;
    g->scratch_live[idx] = true;
    return slot;
}

void plh_gpu_release_scratch(pl_gpu gpu, const void *slot)
{
    struct gpu_priv *g = GPU_PRIV(gpu);
    if (!slot || !g->scratch)
        return;
    const size_t off = (const uint8_t *) slot - (const uint8_t *) g->scratch;
    if (off < (size_t) PLH_SCRATCH_SLOTS * PLH_SCRATCH_BYTES)
        g->scratch_live[off / PLH_SCRATCH_BYTES] = false;
}

bool plh_buf_read(pl_gpu gpu, pl_buf buf, size_t buf_offset, void *dest, size_t size)
{
    struct gpu_priv *g = GPU_PRIV(gpu);
    queued_unnumbered(g);
    int err = plh_copy2d_d2h(g->stream, dest, size, (uint8_t *) BUF_PRIV(buf)->ptr + buf_offset,
                             size, size, 1);
    err = err ? err : sync_main(g);
    return !err;
}

static void hip_buf_copy(pl_gpu gpu, pl_buf dst, size_t dst_offset, pl_buf src, size_t src_offset,
                         size_t size)
{
    queued_unnumbered(GPU_PRIV(gpu));
    plh_copy2d_d2d(GPU_PRIV(gpu)->stream, (uint8_t *) BUF_PRIV(dst)->ptr + dst_offset, size,
                   (uint8_t *) BUF_PRIV(src)->ptr + src_offset, size, size, 1);
}

static bool hip_buf_copy_swap(pl_gpu gpu, pl_buf dst, size_t dst_offset, pl_buf src, size_t src_offset,
                              size_t size, int wordsize)
{
    queued_unnumbered(GPU_PRIV(gpu));
    return !plh_launch_swap_words(GPU_PRIV(gpu)->stream, (const uint8_t *) BUF_PRIV(src)->ptr + src_offset,
                                  (uint8_t *) BUF_PRIV(dst)->ptr + dst_offset, size / 4, wordsize);
}

static bool hip_buf_export(pl_gpu gpu, pl_buf buf)
{
    (void) buf;
    pl_msg(gpu->log, PL_LOG_ERR, "pl_buf_export: buffers of this backend have no exportable handle");
    return false;
}

static bool hip_buf_poll(pl_gpu gpu, pl_buf buf, uint64_t timeout)
{
    (void) buf;
    return stream_busy(gpu, timeout);
}

/* ------------------------------------------------------------------------ */
/* passes: a recorded op list behind the reference's pass interface (gpu.h)   */

struct pass_priv {
    struct pl_pass_t pub;
    struct plh_pass pass;
    bool transpose, detect_peak;
    void *polar_obj;
    pl_shader_obj peak_state;
    int out_w, out_h;
    pl_shader_obj held[16];
    int num_held;
    pl_buf noise;
    char *text;
    struct pl_desc *descs;
};

static pl_pass hip_pass_create(pl_gpu gpu, const struct pl_pass_params *params, pl_shader sh)
{
    (void) gpu;
    struct pass_priv *p = calloc(1, sizeof(*p));
    if (!p)
        return NULL;
    p->pub.params = *params;
    p->text = strdup(params->glsl_shader);
    p->pub.params.glsl_shader = p->text;
    p->pub.params.vertex_shader = NULL;
    p->pub.params.vertex_attribs = NULL;
    p->pub.params.num_vertex_attribs = 0;
    p->pub.params.variables = NULL;
    p->pub.params.constants = NULL;
    p->pub.params.constant_data = NULL;
    p->pub.params.descriptors = NULL;
    if (params->num_descriptors) {
        p->descs = calloc(params->num_descriptors, sizeof(*p->descs));
        if (p->descs) {
            memcpy(p->descs, params->descriptors, params->num_descriptors * sizeof(*p->descs));
            for (int i = 0; i < params->num_descriptors; i++)
                p->descs[i].name = NULL;
        }
        p->pub.params.descriptors = p->descs;
    }
    if (!p->text || (params->num_descriptors && !p->descs)) {
        free(p->text);
        free(p->descs);
        free(p);
        return NULL;
    }
    p->pass = sh->pass;
    p->transpose = sh->transpose;
    p->detect_peak = sh->detect_peak;
    p->peak_state = sh->peak_state;
    p->polar_obj = sh->polar_obj;
    p->out_w = sh->output_w;
    p->out_h = sh->output_h;
    for (int i = 0; i < sh->num_held; i++) {
        p->held[p->num_held++] = sh->held[i];
        sh->held[i]->refcount++;
    }
    return &p->pub;
}

static void hip_pass_destroy(pl_gpu gpu, pl_pass pass)
{
    struct pass_priv *p = (struct pass_priv *) pass;
    hip_gpu_finish(gpu);    // launches of this pass may still read its objects
    for (int i = 0; i < p->num_held; i++)
        pl_shader_obj_destroy(&p->held[i]);
    pl_buf_destroy(gpu, &p->noise);
    free(p->text);
    free(p->descs);
    free(p);
}

static void hip_pass_run(pl_gpu gpu, const struct pl_pass_run_params *params, pl_tex target,
                         pl_rect2d rc)
{
    struct gpu_priv *g = GPU_PRIV(gpu);
    struct pass_priv *p = (struct pass_priv *) params->pass;
    struct plh_pass local = p->pass;    // the stored op list stays as it was recorded
    const struct plh_pass_exec x = {
        .pass = &local, .transpose = p->transpose, .polar_obj = p->polar_obj,
        .detect_peak = p->detect_peak, .peak_state = p->peak_state,
    };
    int err;
    if (target) {
        const int tw = abs(pl_rect_w(rc)), th = abs(pl_rect_h(rc));
        const int need_w = p->transpose ? p->out_h : p->out_w, need_h = p->transpose ? p->out_w : p->out_h;
        if (need_w && need_h && (need_w != tw || need_h != th)) {
            pl_msg(gpu->log, PL_LOG_ERR, "pl_pass_run: the pass was recorded for a %dx%d output, "
                   "the target rect is %dx%d", need_w, need_h, tw, th);
            return;
        }
        err = plh_pass_execute(gpu, gpu->log, &x, target, rc, params->timer, &p->noise);
    } else {
        // a pass without an image output (a measurement): it covers its recorded output size
        if (!p->out_w || !p->out_h) {
            pl_msg(gpu->log, PL_LOG_ERR, "pl_pass_run: a compute pass without a storage image "
                   "needs a shader with a defined output size");
            return;
        }
        memset(&local.dst, 0, sizeof(local.dst));
        local.width = p->out_w;
        local.height = p->out_h;
        local.out_scale[0] = 1.0 / p->out_w;
        local.out_scale[1] = 1.0 / p->out_h;
        local.base_x = local.base_y = 0;
        local.dir_x = local.dir_y = 1;
        local.transpose = 0;
        local.frag_x0 = local.frag_y0 = 0;
        if (params->timer)
            plh_timer_begin(gpu, params->timer, 0);
        queued_unnumbered(g);
        err = plh_launch_pass(g->stream, &local);
        if (params->timer)
            plh_timer_end(gpu, params->timer, 0);
        if (!err && p->detect_peak)
            plh_peak_pass_launched(gpu, p->peak_state, 0, 0, false);
    }
    if (err) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_pass_run: %s", plh_strerror(err));
        g->failed = true;
    }
}

/* ------------------------------------------------------------------------ */
/* timers: ring of hipEvent pairs                                            */

static void hip_timer_destroy(pl_gpu gpu, pl_timer t)
{
    (void) gpu;
    for (int i = 0; i < PLH_TIMER_RING; i++) {
        plh_event_destroy(t->start[i]);
        plh_event_destroy(t->stop[i]);
    }
    free(t);
}

static pl_timer hip_timer_create(pl_gpu gpu)
{
    struct pl_timer_t *t = calloc(1, sizeof(*t));
    if (!t)
        return NULL;
    for (int i = 0; i < PLH_TIMER_RING; i++) {
        if (plh_event_create(&t->start[i]) || plh_event_create(&t->stop[i])) {
            hip_timer_destroy(gpu, t);
            return NULL;
        }
    }
    return t;
}

void plh_timer_begin(pl_gpu gpu, pl_timer t, int on)
{
    if (t->head - t->tail >= PLH_TIMER_RING)
        t->tail++; // drop the oldest sample
    plh_event_record(t->start[t->head % PLH_TIMER_RING], plh_gpu_stream_n(gpu, on));
}

void plh_timer_end(pl_gpu gpu, pl_timer t, int on)
{
    plh_event_record(t->stop[t->head % PLH_TIMER_RING], plh_gpu_stream_n(gpu, on));
    t->head++;
}

static uint64_t hip_timer_query(pl_gpu gpu, pl_timer t)
{
    (void) gpu;
    if (t->tail == t->head)
        return 0;
    const int i = t->tail % PLH_TIMER_RING;
    if (plh_event_query(t->stop[i]) != 1)
        return 0;
    uint64_t ns = 0;
    plh_event_elapsed_ns(t->start[i], t->stop[i], &ns);
    t->tail++;
    return PL_MAX(ns, 1);
}

static const struct plh_gpu_fns hip_fns = {
    .destroy        = hip_destroy,
    .tex_create     = hip_tex_create,
    .tex_destroy    = hip_tex_destroy,
    .tex_invalidate = hip_tex_invalidate,
    .tex_clear_ex   = hip_tex_clear_ex,
    .tex_blit       = hip_tex_blit,
    .tex_upload     = hip_tex_upload,
    .tex_download   = hip_tex_download,
    .tex_poll       = hip_tex_poll,
    .buf_create     = hip_buf_create,
    .buf_destroy    = hip_buf_destroy,
    .buf_write      = plh_buf_write,
    .buf_read       = plh_buf_read,
    .buf_copy       = hip_buf_copy,
    .buf_copy_swap  = hip_buf_copy_swap,
    .buf_export     = hip_buf_export,
    .buf_poll       = hip_buf_poll,
    .pass_create    = hip_pass_create,
    .pass_destroy   = hip_pass_destroy,
    .pass_run       = hip_pass_run,
    .timer_create   = hip_timer_create,
    .timer_destroy  = hip_timer_destroy,
    .timer_query    = hip_timer_query,
    .gpu_flush      = hip_gpu_flush,
    .gpu_finish     = hip_gpu_finish,
    .gpu_is_failed  = hip_gpu_is_failed,
};
