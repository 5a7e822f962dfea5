/*
 * libplacebo-hip — pl_dispatch: lowers a finished pl_shader to kernel launches.
 *
 * Counterpart of the reference's src/dispatch.c. The reference finalises GLSL,
 * hashes it, compiles/caches a pl_pass and binds uniforms (finalize_pass
 * :732-970); here the shader already *is* the launch description, so
 * pl_dispatch_finish only has to add the target half:
 *   - rect defaulting / validation                         dispatch.c:1199-1232
 *   - compute-emulated rasterisation: out_scale, base, dir dispatch.c:1028-1142
 *   - blending is not supported (never used on the pl_render_image path with
 *     blend_params == NULL)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <libplacebo/dispatch.h>

#include "shaders_priv.h"

#define MAX_SHADERS 32

struct pass_timing {
    uint64_t signature;
    pl_timer timer;
    struct pl_dispatch_info info;
    int ring;
};

struct pl_dispatch_t {
    pl_log log;
    pl_gpu gpu;
    uint8_t current_ident;
    uint8_t current_index;

    pl_shader pool[MAX_SHADERS];
    int num_pool;

    void (*info_cb)(void *priv, const struct pl_dispatch_info *);
    void *info_priv;
    struct pass_timing timings[64];
    int num_timings;
    pl_buf noise;               // white-noise dither plane of the pass being launched
    // Binned overlays, resident on the device: subtitles and on-screen displays stay the same for
    // seconds, so an overlay that was drawn before (same parts on a target of the same size) is
    // neither binned nor uploaded again -- an upload is a copy engine's turn on the stream between
    // two kernels, which costs the frame more than the overlay's kernel does
    struct overlay_layout {
        int num_parts, w, h;
        struct plh_overlay_part *parts;     // host copy: what the entry was built from
        pl_buf buf;                         // parts | tiles | order
        size_t parts_bytes, tiles_bytes;
        uint32_t num_tiles;
        uint64_t used;                      // dp->overlay_clock of the last use
    } overlays[8];
    uint64_t overlay_clock;
    pl_tex blend_tmp;           // a pass with blend_params renders here first (rgba32f)
    uint32_t *bins;             // host scratch of plh_dispatch_overlay
    size_t bins_cap;
    uint8_t *blob;
    size_t blob_cap;
};

pl_dispatch pl_dispatch_create(pl_log log, pl_gpu gpu)
{
    struct pl_dispatch_t *dp = calloc(1, sizeof(*dp));
    if (!dp)
        return NULL;
    dp->log = log;
    dp->gpu = gpu;
    return dp;
}

void pl_dispatch_destroy(pl_dispatch *ptr)
{
    pl_dispatch dp = ptr ? *ptr : NULL;
    if (!dp)
        return;
    for (int i = 0; i < dp->num_pool; i++)
        pl_shader_free(&dp->pool[i]);
    pl_buf_destroy(dp->gpu, &dp->noise);
    for (size_t i = 0; i < PL_ARRAY_SIZE(dp->overlays); i++) {
        pl_buf_destroy(dp->gpu, &dp->overlays[i].buf);
        free(dp->overlays[i].parts);
    }
    pl_tex_destroy(dp->gpu, &dp->blend_tmp);
    free(dp->bins);
    free(dp->blob);
    for (int i = 0; i < dp->num_timings; i++) {
        pl_timer_destroy(dp->gpu, &dp->timings[i].timer);
        pl_shader_info_deref(&dp->timings[i].info.shader);
    }
    free(dp);
    *ptr = NULL;
}

void pl_dispatch_reset_frame(pl_dispatch dp)
{
    dp->current_ident = 0;
    dp->current_index++; // uint8 wrap-around, like dispatch.c:1618
}

pl_shader pl_dispatch_begin(pl_dispatch dp)
{
    struct pl_shader_params params = {
        .id = dp->current_ident++,
        .gpu = dp->gpu,
        .index = dp->current_index,
    };
    if (dp->num_pool) {
        pl_shader sh = dp->pool[--dp->num_pool];
        pl_shader_reset(sh, &params);
        return sh;
    }
    return pl_shader_alloc(dp->log, &params);
}

void pl_dispatch_abort(pl_dispatch dp, pl_shader *psh)
{
    pl_shader sh = psh ? *psh : NULL;
    if (!sh)
        return;
    // release held objects right away, keep the allocation for reuse
    pl_shader_reset(sh, NULL);
    if (dp->num_pool < MAX_SHADERS) {
        dp->pool[dp->num_pool++] = sh;
    } else {
        pl_shader_free(&sh);
    }
    *psh = NULL;
}

void pl_dispatch_callback(pl_dispatch dp, void *priv,
                          void (*cb)(void *priv, const struct pl_dispatch_info *))
{
    dp->info_cb = cb;
    dp->info_priv = priv;
}

static uint64_t hash_str(const char *s)
{
    uint64_t h = 1469598103934665603ull; // FNV-1a
    for (; *s; s++)
        h = (h ^ (uint8_t) *s) * 1099511628211ull;
    return h;
}

static struct pass_timing *get_timing(pl_dispatch dp, pl_shader sh)
{
    if (!dp->info_cb)
        return NULL;
    const struct pl_shader_res *res = pl_shader_finalize(sh);
    if (!res)
        return NULL;
    // keyed by what the pass does, not by its per-frame values (PRNG seeds, LUT contents)
    const uint64_t sig = hash_str(res->info->description) ^
                         ((uint64_t) sh->pass.num_ops << 56) ^
                         ((uint64_t) sh->pass.s.type << 48);
    for (int i = 0; i < dp->num_timings; i++) {
        if (dp->timings[i].signature == sig)
            return &dp->timings[i];
    }
    if (dp->num_timings == (int) PL_ARRAY_SIZE(dp->timings))
        return NULL;
    struct pass_timing *t = &dp->timings[dp->num_timings++];
    memset(t, 0, sizeof(*t));
    t->signature = sig;
    t->timer = pl_timer_create(dp->gpu);
    t->info.shader = pl_shader_info_ref(res->info);
    t->info.signature = sig;
    return t;
}

static void drain_timing(pl_dispatch dp, struct pass_timing *t)
{
    if (!t || !t->timer)
        return;
    for (uint64_t ns; (ns = pl_timer_query(dp->gpu, t->timer));) {
        struct pl_dispatch_info *info = &t->info;
        info->samples[t->ring] = ns;
        t->ring = (t->ring + 1) % (int) PL_ARRAY_SIZE(info->samples);
        info->num_samples = PL_MIN(info->num_samples + 1, (int) PL_ARRAY_SIZE(info->samples));
        info->last = ns;
        info->peak = PL_MAX(info->peak, ns);
        uint64_t sum = 0;
        for (int i = 0; i < info->num_samples; i++)
            sum += info->samples[i];
        info->average = sum / info->num_samples;
        dp->info_cb(dp->info_priv, info);
    }
}

// k_pass lanes own 2x2 output cells. For a bilinear upscale, pick the cell phase for which
// the outputs of a cell share their 2x2 source footprint (an efficiency hint only: the kernel
// checks the footprints itself).
static void plh_pass_choose_cells(struct plh_pass *pass)
{
    pass->cell_padx = pass->cell_pady = 0;
    // Unorm targets are final frames / output planes (intermediate FBOs are float): nothing on
    // this GPU reads them again, so their stores bypass the caches. PL_HIP_NT_STORE=0 disables.
    const char *nt_env = getenv("PL_HIP_NT_STORE");
    pass->nt_store = (!nt_env || atoi(nt_env)) && pass->dst.fmt <= PLH_FMT_RGBA16;
    // extra planes (PLANE_FETCH): the same identity-fetch rule as for the main sampler
    for (int i = 0; i < pass->num_ops; i++) {
        struct plh_op *op = &pass->ops[i];
        if (op->kind != PLH_OP_PLANE_FETCH || !((op->i2 >> 12) & 1) || !((op->i2 >> 15) & 1))
            continue;
        if (fabsf(pass->width / op->f[9] - 1.0f) < 1e-6f &&
            fabsf(pass->height / op->f[10] - 1.0f) < 1e-6f)
            op->i2 &= ~(1 << 12);
    }
    if (pass->s.type != PLH_SAMPLE_BILINEAR)
        return;
    // identity fetch (1:1, on the texel grid): what a texture unit returns is the texel
    if (pass->s.rect_on_grid && fabsf(pass->width / pass->s.rect_w - 1.0f) < 1e-6f &&
        fabsf(pass->height / pass->s.rect_h - 1.0f) < 1e-6f)
    {
        pass->s.type = PLH_SAMPLE_NEAREST;
        return;
    }
    const struct plh_sampler_args *s = &pass->s;
    for (int axis = 0; axis < 2; axis++) {
        const int n = axis ? pass->height : pass->width;
        if (n < 2)
            continue;
        // texel-space coordinate of the first two outputs along this axis
        const double p0 = s->pos[0][axis], p1 = axis ? s->pos[2][axis] : s->pos[1][axis];
        const double size = axis ? s->src.h : s->src.w;
        const double u0 = (p0 + (p1 - p0) * (0.5 / n)) * size - 0.5;
        const double u1 = (p0 + (p1 - p0) * (1.5 / n)) * size - 0.5;
        const int pad = floor(u0) != floor(u1);
        if (axis)
            pass->cell_pady = pad;
        else
            pass->cell_padx = pad;
    }
}

// PL_DITHER_WHITE_NOISE is recorded as (i1 = 2, i0 = seed). The kernels only know dither
// matrices: now that the size of the pass is known, evaluate the PRNG for its fragment
// coordinates into a plane whose row stride is a power of two (the matrix lookup wraps x and y
// with one mask) and turn the op into a plain LUT dither over it (k_noise.hip says why). Only the
// width x height corner carries noise, but the WHOLE square is allocated (16 MiB for a 1080p pass,
// 64 MiB for 4K, and only while this method is selected): every kernel fetches the dither value
// of the lanes its tiles pad the rect with BEFORE the store guard drops them, and those lanes'
// coordinates go through the same mask -- rows up to a tile beyond the height, and row / column
// -1 of the padded first cell, which the mask wraps to side - 1. (A plane of `height` rows, and
// then one rounded up to a multiple of 256, both faulted on a 960 x 402 polar pass:
// tests/test_gpu_dither.py::test_white_noise_plane_covers_the_padded_rows.)
static bool realize_white_noise(pl_gpu gpu, plh_stream stream, pl_buf *noise, struct plh_pass *pass)
{
    for (int i = 0; i < pass->num_ops; i++) {
        struct plh_op *op = &pass->ops[i];
        if (op->kind != PLH_OP_DITHER || op->i1 != 2)
            continue;
        int side = 16;
        while (side < pass->width || side < pass->height)
            side <<= 1;
        const size_t size = (size_t) side * side * sizeof(float);
        if (!*noise || (*noise)->params.size < size) {
            pl_buf_destroy(gpu, noise);
            *noise = pl_buf_create(gpu, pl_buf_params(.size = size, .storable = true));
            if (!*noise)
                return false;
        }
        if (plh_launch_white_noise(stream, pl_hip_buf_ptr(*noise), side,
                                   pass->width, pass->height, pass->frag_x0, pass->frag_y0,
                                   (uint32_t) op->i0))
            return false;
        op->i0 = side;
        op->i1 = 0;     // LUT
        op->i2 = 0;     // not rotated
        op->f[2] = 1.0f / side;
        op->ptr = pl_hip_buf_ptr(*noise);
        op->ptr2 = NULL;    // (no transposed copy of a noise plane)
    }
    return true;
}

// The target half of a pass and its launch: shared by pl_dispatch_finish and pl_pass_run (gpu.c).
int plh_pass_execute(pl_gpu gpu, pl_log log, const struct plh_pass_exec *x, pl_tex target,
                     pl_rect2d rc, pl_timer timer, pl_buf *noise)
{
    struct plh_pass *pass = x->pass;
    const int tw = abs(pl_rect_w(rc)), th = abs(pl_rect_h(rc));
    plh_tex_view(target, &pass->dst);
    pass->width = x->transpose ? th : tw;
    pass->height = x->transpose ? tw : th;
    pass->out_scale[0] = 1.0 / pass->width;
    pass->out_scale[1] = 1.0 / pass->height;
    pass->base_x = rc.x0 - (rc.x0 > rc.x1);
    pass->base_y = rc.y0 - (rc.y0 > rc.y1);
    pass->dir_x = rc.x0 > rc.x1 ? -1 : 1;
    pass->dir_y = rc.y0 > rc.y1 ? -1 : 1;
    pass->transpose = x->transpose;
    pass->frag_x0 = pass->frag_y0 = 0; // compute passes: rect-relative gl_FragCoord
    // ... except for the tile pattern under a transparent image, which must continue the border's
    // (pl_frame_clear_tiles: absolute plane coordinates; the reference's output pass is a raster
    // pass whose gl_FragCoord is the target pixel, src/renderer.c:2745)
    for (int i = 0; i < pass->num_ops; i++) {
        struct plh_op *op = &pass->ops[i];
        if (op->kind != PLH_OP_BLEND_TILES)
            continue;
        op->f[9] = pass->dir_x;
        op->f[10] = pass->base_x + 0.5f - 0.5f * pass->dir_x;
        op->f[11] = pass->dir_y;
        op->f[12] = pass->base_y + 0.5f - 0.5f * pass->dir_y;
    }

    if (pass->s.type == PLH_SAMPLE_POLAR && x->polar_obj)
        plh_polar_pp_setup(gpu, log, x->polar_obj, pass);
    plh_pass_choose_cells(pass);
    // which stream (gpu_hip.c "two streams"); every call here is a no-op in one-stream mode
    const int on = x->on_aux && plh_gpu_async(gpu) && !plh_gpu_has_peak_exchange(gpu);
    if (!realize_white_noise(gpu, plh_gpu_stream_n(gpu, on), noise, pass))
        return -1005;
    if (on)
        plh_gpu_order_after(gpu, 1, x->aux_after);
    const uint64_t seq = plh_tex_order(gpu, on, x->src_tex, target);
    if (timer)
        plh_timer_begin(gpu, timer, on);
    // The event that marks the end of this pass -- the measurement's `written` event, or a fence
    // of the two-stream bookkeeping -- rides on the launch of the pass's last kernel where the
    // launcher takes it; recorded behind the pass it is a queue entry that holds the stream's
    // next kernel back (backend.h: plh_launch_offer_stop).
    plh_event stop = x->detect_peak ? plh_peak_written_event(x->peak_state)
                                    : plh_gpu_fence_for_launch(gpu, on, x->src_tex);
    plh_launch_offer_stop(stop);
    const int err = plh_launch_pass(plh_gpu_stream_n(gpu, on), pass);
    const bool taken = plh_launch_stop_taken() && !err;
    if (timer)
        plh_timer_end(gpu, timer, on);
    if (!err && x->detect_peak)
        plh_peak_pass_launched(gpu, x->peak_state, on, seq, taken);
    else if (stop)
        plh_gpu_fence_launched(gpu, on, taken);
    return err;
}

/* ---- overlays and blended stores (k_overlay.hip) --------------------------------------------- */

static int blend_factor(enum pl_blend_mode m)
{
    switch (m) {
    case PL_BLEND_ZERO:                return PLH_BLEND_ZERO;
    case PL_BLEND_ONE:                 return PLH_BLEND_ONE;
    case PL_BLEND_SRC_ALPHA:           return PLH_BLEND_SRC_ALPHA;
    case PL_BLEND_ONE_MINUS_SRC_ALPHA: return PLH_BLEND_ONE_MINUS_SRC_ALPHA;
    case PL_BLEND_MODE_COUNT:          break;
    }
    return PLH_BLEND_ZERO;
}

// the target pixels whose centre lies in [a, b): first and one past the last, within [0, n)
static void covered_range(float a, float b, int n, int *first, int *end)
{
    if (!(b > a)) {     // empty, or not a number: covers nothing (the kernel's test says the same)
        *first = *end = 0;
        return;
    }
    const float lo = ceilf(a - 0.5f), hi = ceilf(b - 0.5f);
    *first = lo <= 0.0f ? 0 : lo >= (float) n ? n : (int) lo;
    *end = hi <= 0.0f ? 0 : hi >= (float) n ? n : (int) hi;
}

static bool reserve(void **ptr, size_t *cap, size_t need)
{
    if (need <= *cap)
        return true;
    void *grown = realloc(*ptr, need);
    if (!grown)
        return false;
    *ptr = grown;
    *cap = need;
    return true;
}

// The rasteriser's work, done on the host: every part is clipped to the w x h target and entered
// into the list of each 16 x 16 tile it touches; tiles keep their parts in drawing order. The
// result goes to the device as one blob (parts | tiles {xy, first, count} | order). Returns the
// cache entry that holds it (num_tiles == 0: nothing of the overlay is visible), NULL on failure.
static struct overlay_layout *overlay_layout(pl_dispatch dp, const struct plh_overlay_part *parts,
                                             int n, int w, int h)
{
    const size_t parts_bytes = (size_t) n * sizeof(struct plh_overlay_part);
    struct overlay_layout *lay = &dp->overlays[0];
    for (size_t i = 0; i < PL_ARRAY_SIZE(dp->overlays); i++) {
        struct overlay_layout *e = &dp->overlays[i];
        if (e->parts && e->num_parts == n && e->w == w && e->h == h &&
            !memcmp(e->parts, parts, parts_bytes))
        {
            e->used = ++dp->overlay_clock;
            return e;
        }
        if (e->used < lay->used)
            lay = e;    // the least recently used one is replaced (its buffer is overwritten by
                        // copies on the stream its kernels ran on: behind them)
    }

    const int T = PLH_OVERLAY_TILE;
    const int tiles_w = (w + T - 1) / T, tiles_h = (h + T - 1) / T;
    const size_t num_bins = (size_t) tiles_w * tiles_h;
    if (!reserve((void **) &dp->bins, &dp->bins_cap, 2 * num_bins * sizeof(uint32_t)))
        return NULL;
    uint32_t *count = dp->bins, *start = dp->bins + num_bins;
    memset(count, 0, num_bins * sizeof(uint32_t));

    // pass 1: how many parts every tile holds
    size_t total = 0;
    for (int i = 0; i < n; i++) {
        int xs, xe, ys, ye;
        covered_range(parts[i].x0, parts[i].x1, w, &xs, &xe);
        covered_range(parts[i].y0, parts[i].y1, h, &ys, &ye);
        if (xs >= xe || ys >= ye)
            continue;
        for (int ty = ys / T; ty <= (ye - 1) / T; ty++) {
            for (int tx = xs / T; tx <= (xe - 1) / T; tx++) {
                count[(size_t) ty * tiles_w + tx]++;
                total++;
            }
        }
    }
    size_t num_tiles = 0;
    for (size_t b = 0; b < num_bins; b++)
        num_tiles += count[b] != 0;

    const size_t tiles_bytes = num_tiles * 3 * sizeof(uint32_t);
    const size_t blob_bytes = parts_bytes + tiles_bytes + total * sizeof(uint32_t);
    struct plh_overlay_part *copy = realloc(lay->parts, PL_MAX(parts_bytes, 1));
    if (copy)
        lay->parts = copy;  // (on failure the old block stays the entry's, to be reused or freed)
    lay->num_parts = -1;    // the entry no longer describes what its buffer holds
    if (!copy || !reserve((void **) &dp->blob, &dp->blob_cap, blob_bytes))
        return NULL;
    memcpy(copy, parts, parts_bytes);
    *lay = (struct overlay_layout) {
        .num_parts = n, .w = w, .h = h, .parts = copy, .buf = lay->buf,
        .parts_bytes = parts_bytes, .tiles_bytes = tiles_bytes, .num_tiles = num_tiles,
        .used = ++dp->overlay_clock,
    };
    if (!num_tiles)
        return lay;

    memcpy(dp->blob, parts, parts_bytes);
    uint32_t *tiles = (uint32_t *) (dp->blob + parts_bytes);
    uint32_t *order = (uint32_t *) (dp->blob + parts_bytes + tiles_bytes);
    uint32_t next = 0, t = 0;
    for (size_t b = 0; b < num_bins; b++) {
        if (!count[b])
            continue;
        tiles[3 * t + 0] = (uint32_t) (b % tiles_w) | (uint32_t) (b / tiles_w) << 16;
        tiles[3 * t + 1] = next;
        tiles[3 * t + 2] = count[b];
        start[b] = next;
        next += count[b];
        t++;
    }
    // pass 2: the parts of every tile, in drawing order
    for (int i = 0; i < n; i++) {
        int xs, xe, ys, ye;
        covered_range(parts[i].x0, parts[i].x1, w, &xs, &xe);
        covered_range(parts[i].y0, parts[i].y1, h, &ys, &ye);
        if (xs >= xe || ys >= ye)
            continue;
        for (int ty = ys / T; ty <= (ye - 1) / T; ty++) {
            for (int tx = xs / T; tx <= (xe - 1) / T; tx++)
                order[start[(size_t) ty * tiles_w + tx]++] = i;
        }
    }

    if (!lay->buf || lay->buf->params.size < blob_bytes) {
        pl_buf_destroy(dp->gpu, &lay->buf);
        lay->buf = pl_buf_create(dp->gpu, pl_buf_params(
            .size = PL_MAX(blob_bytes * 2, (size_t) 16 * 1024), .storable = true));
        if (!lay->buf) {
            lay->num_parts = -1;
            return NULL;
        }
    }
    // (in pieces the staging slots take: a larger write waits for the stream)
    for (size_t off = 0; off < blob_bytes; off += PLH_STAGE_BYTES) {
        plh_buf_write(dp->gpu, lay->buf, off, dp->blob + off,
                      PL_MIN(blob_bytes - off, (size_t) PLH_STAGE_BYTES));
    }
    return lay;
}

// Draw `draw->parts`, in order, into `target` (the reference's pl_dispatch_vertex call of
// draw_overlays, src/renderer.c:1004-1019; and the blending half of pl_dispatch_finish)
bool plh_dispatch_overlay(pl_dispatch dp, pl_shader *psh, pl_tex target,
                          const struct plh_overlay_draw *draw)
{
    pl_shader sh = *psh;
    bool ok = false;
    if (sh->failed) {
        pl_msg(dp->log, PL_LOG_ERR, "Trying to dispatch a failed shader.");
        goto done;
    }
    if (sh->kind != PLH_SHADER_PASS || sh->pass.s.type != PLH_SAMPLE_NONE ||
        sh->output != PL_SHADER_SIG_COLOR)
    {
        pl_msg(dp->log, PL_LOG_ERR, "An overlay shader consists of colour operations only");
        goto done;
    }
    if (!target->params.storable || !draw->tex) {
        pl_msg(dp->log, PL_LOG_ERR, "Trying to draw an overlay using an invalid target or texture");
        goto done;
    }

    const struct overlay_layout *lay = overlay_layout(dp, draw->parts, draw->num_parts,
                                                      target->params.w, target->params.h);
    if (!lay) {
        pl_msg(dp->log, PL_LOG_ERR, "Failed placing an overlay of %d parts", draw->num_parts);
        goto done;
    }
    if (!lay->num_tiles) {
        ok = true;      // nothing of it is visible
        goto done;
    }

    struct plh_pass *pass = &sh->pass;
    plh_tex_view(draw->tex, &pass->s.src);
    plh_tex_view(target, &pass->dst);
    pass->num_pre_ops = PL_MIN(PL_MAX(draw->coverage_at, 0), pass->num_ops);
    const uint8_t *dev = pl_hip_buf_ptr(lay->buf);
    const struct pl_blend_params *bl = draw->blend;
    const struct plh_overlay_args args = {
        .parts = (const struct plh_overlay_part *) dev,
        .tiles = (const uint32_t *) (dev + lay->parts_bytes),
        .order = (const uint32_t *) (dev + lay->parts_bytes + lay->tiles_bytes),
        .num_tiles = (int32_t) lay->num_tiles,
        .mode = draw->mode,
        .linear = draw->linear,
        .premultiplied = draw->premultiplied,
        .blend = bl != NULL,
        .src_rgb = bl ? blend_factor(bl->src_rgb) : PLH_BLEND_ONE,
        .dst_rgb = bl ? blend_factor(bl->dst_rgb) : PLH_BLEND_ZERO,
        .src_alpha = bl ? blend_factor(bl->src_alpha) : PLH_BLEND_ONE,
        .dst_alpha = bl ? blend_factor(bl->dst_alpha) : PLH_BLEND_ZERO,
    };
    struct pass_timing *timing = get_timing(dp, sh);
    plh_tex_order(dp->gpu, 0, draw->tex, target);
    if (timing)
        plh_timer_begin(dp->gpu, timing->timer, 0);
    const int err = plh_launch_overlay(plh_gpu_stream(dp->gpu), pass, &args);
    if (timing)
        plh_timer_end(dp->gpu, timing->timer, 0);
    if (err) {
        pl_msg(dp->log, PL_LOG_ERR, "Failed launching overlay pass: %s", plh_strerror(err));
        goto done;
    }
    drain_timing(dp, timing);
    ok = true;

done:
    pl_dispatch_abort(dp, psh);
    return ok;
}

// pl_dispatch_params.blend_params: the pass renders into an rgba32f image of the rect's size
// (what the fragment shader would have handed to the blend unit, unrounded), which is then
// blended into the target texel by texel
static bool finish_blended(pl_dispatch dp, pl_shader sh, const struct plh_pass_exec *x,
                           pl_tex target, pl_rect2d rc, pl_timer timer,
                           const struct pl_blend_params *blend)
{
    const int tw = abs(pl_rect_w(rc)), th = abs(pl_rect_h(rc));
    pl_fmt fmt = pl_find_named_fmt(dp->gpu, "rgba32f");
    if (!fmt || !pl_tex_recreate(dp->gpu, &dp->blend_tmp, pl_tex_params(
            .w = tw, .h = th, .format = fmt, .sampleable = true, .storable = true)))
    {
        pl_msg(dp->log, PL_LOG_ERR, "Failed creating the intermediate image of a blended pass");
        return false;
    }
    const pl_rect2d local = {
        .x0 = rc.x0 > rc.x1 ? tw : 0, .x1 = rc.x0 > rc.x1 ? 0 : tw,
        .y0 = rc.y0 > rc.y1 ? th : 0, .y1 = rc.y0 > rc.y1 ? 0 : th,
    };
    struct plh_pass_exec unblended = *x;
    unblended.on_aux = false;
    const int err = plh_pass_execute(dp->gpu, dp->log, &unblended, dp->blend_tmp, local, timer,
                                     &dp->noise);
    if (err) {
        pl_msg(dp->log, PL_LOG_ERR, "Failed launching pass '%s': %s",
               sh_description(sh), plh_strerror(err));
        return false;
    }
    const struct plh_overlay_part whole = {
        .x0 = PL_MIN(rc.x0, rc.x1), .y0 = PL_MIN(rc.y0, rc.y1),
        .x1 = PL_MAX(rc.x0, rc.x1), .y1 = PL_MAX(rc.y0, rc.y1),
    };
    pl_shader copy = pl_dispatch_begin(dp);
    if (!copy)
        return false;
    copy->output = PL_SHADER_SIG_COLOR;
    return plh_dispatch_overlay(dp, &copy, target, &(struct plh_overlay_draw) {
        .tex = dp->blend_tmp, .mode = PLH_OVERLAY_TEXEL, .blend = blend,
        .parts = &whole, .num_parts = 1,
    });
}

// Deprecated front ends of the gpu's pl_cache (src/dispatch.c:1624-1632)
/* ---- the two passes of a separable downscale as one launch (k_lowpass2, k_ortho.hip) ---------- */

// `vert`: the vertical pass of the contrast-recovery low-pass, recorded (pl_shader_sample_ortho2) on
// the full-size r16hf plane but not dispatched; `horiz`: the horizontal one, recorded on a texture of
// the intermediate's size that is never written. Returns 1 = launched as one kernel into `target`
// (both shaders consumed), 0 = not a shape the fused kernel takes (nothing consumed: dispatch them
// one after the other), -1 = failed (both consumed).
int plh_dispatch_lowpass2(pl_dispatch dp, pl_shader *pvert, pl_shader *phoriz, pl_tex target)
{
    pl_shader v = *pvert, h = *phoriz;
    int mw, mh, ow, oh;
    if (v->failed || h->failed || v->kind != PLH_SHADER_PASS || h->kind != PLH_SHADER_PASS ||
        v->transpose || h->transpose || v->detect_peak || h->detect_peak || v->on_aux || h->on_aux ||
        !pl_shader_output_size(v, &mw, &mh) || !pl_shader_output_size(h, &ow, &oh))
        return 0;
    const struct plh_pass *pv = &v->pass, *ph = &h->pass;
    const struct plh_sampler_args *sv = &pv->s, *sh_ = &ph->s;
    if (sv->type != PLH_SAMPLE_ORTHO || sh_->type != PLH_SAMPLE_ORTHO || sv->dir != 1 || sh_->dir != 0 ||
        pv->num_ops || ph->num_ops || sv->use_linear != sh_->use_linear || sv->use_ar || sh_->use_ar ||
        sv->linear || sh_->linear || (sv->comp_mask & 1u) != 1u || (sh_->comp_mask & 1u) != 1u ||
        sv->address_mode != sh_->address_mode ||
        (sv->address_mode != PLH_ADDRESS_MIRROR && sv->address_mode != PLH_ADDRESS_CLAMP))
        return 0;
    if (!target || !target->params.storable || target->params.w != ow || target->params.h != oh ||
        oh != mh || sh_->src.w != mw || sh_->src.h != mh || sv->src.w != mw)
        return 0;

    struct plh_lowpass2 a = {0};
    a.src = sv->src;
    plh_tex_view(target, &a.dst);
    memcpy(a.pos_v, sv->pos, sizeof(a.pos_v));
    memcpy(a.pos_h, sh_->pos, sizeof(a.pos_h));
    // (plh_pass_execute's out_scale of either pass)
    a.os_v[0] = 1.0 / mw; a.os_v[1] = 1.0 / mh;
    a.os_h[0] = 1.0 / ow; a.os_h[1] = 1.0 / oh;
    a.mid_w = mw; a.mid_h = mh;
    a.wgt_v = sv->weights; a.wgt_h = sh_->weights;
    a.n_v = sv->row_size; a.stride_v = sv->row_stride;
    a.n_h = sh_->row_size; a.stride_h = sh_->row_stride;
    a.scale_v = sv->scale; a.scale_h = sh_->scale;
    a.mirror = sv->address_mode == PLH_ADDRESS_MIRROR;
    a.linear_trick = sv->use_linear;
    // the tile of a 32 x 16 workgroup: the taps of its first and last output are (n - 1) x ratio
    // texels apart, + the tap count, + one texel of slack for the per-pixel rounding of the geometry
    const double rx = fabs((sh_->pos[1][0] - sh_->pos[0][0]) * mw) / ow;
    const double ry = fabs((sv->pos[2][1] - sv->pos[0][1]) * sv->src.h) / mh;
    a.cols_cap = ((int) ceil(31 * rx) + a.n_h + 3 + 3) & ~3;
    a.rows_cap = (int) ceil(15 * ry) + a.n_v + 3;

    if (!plh_lowpass2_applies(&a))
        return 0;

    pl_gpu gpu = dp->gpu;
    struct pass_timing *timing = get_timing(dp, h);
    pl_timer timer = timing ? timing->timer : NULL;
    const uint64_t seq = plh_tex_order(gpu, 0, v->src_tex, target);
    (void) seq;
    if (timer)
        plh_timer_begin(gpu, timer, 0);
    plh_event stop = plh_gpu_fence_for_launch(gpu, 0, v->src_tex);
    plh_launch_offer_stop(stop);
    const int err = plh_launch_lowpass2(plh_gpu_stream_n(gpu, 0), &a);
    const bool taken = plh_launch_stop_taken() && !err;
    if (timer)
        plh_timer_end(gpu, timer, 0);
    if (stop)
        plh_gpu_fence_launched(gpu, 0, taken);
    drain_timing(dp, timing);
    pl_dispatch_abort(dp, pvert);
    pl_dispatch_abort(dp, phoriz);
    if (err) {
        pl_msg(dp->log, PL_LOG_ERR, "Failed launching the fused low-pass: %s", plh_strerror(err));
        return -1;
    }
    return 1;
}

size_t pl_dispatch_save(pl_dispatch dp, uint8_t *out)
{
    return pl_cache_save(plh_gpu_cache(dp->gpu), out, out ? SIZE_MAX : 0);
}

void pl_dispatch_load(pl_dispatch dp, const uint8_t *cache)
{
    pl_cache_load(plh_gpu_cache(dp->gpu), cache, SIZE_MAX);
}

bool pl_dispatch_finish(pl_dispatch dp, const struct pl_dispatch_params *params)
{
    pl_shader sh = *params->shader;
    bool ok = false;

    if (sh->failed) {
        pl_msg(dp->log, PL_LOG_ERR, "Trying to dispatch a failed shader.");
        goto done;
    }
    if (sh->input != PL_SHADER_SIG_NONE || sh->output != PL_SHADER_SIG_COLOR) {
        pl_msg(dp->log, PL_LOG_ERR, "Trying to dispatch shader with incompatible signature!");
        goto done;
    }
    if (sh->kind != PLH_SHADER_PASS) {
        pl_msg(dp->log, PL_LOG_ERR, "This shader must be run with pl_dispatch_compute");
        goto done;
    }
    pl_tex target = params->target;
    if (!target || pl_tex_params_dimension(target->params) != 2) {
        pl_msg(dp->log, PL_LOG_ERR, "Trying to dispatch a shader using an invalid target");
        goto done;
    }
    if (!target->params.storable) {
        pl_msg(dp->log, PL_LOG_ERR, "Trying to dispatch using a non-storable target "
               "(every pass is a compute pass on this backend).");
        goto done;
    }

    pl_rect2d rc = params->rect;
    if (!pl_rect_w(rc)) {
        rc.x0 = 0;
        rc.x1 = target->params.w;
    }
    if (!pl_rect_h(rc)) {
        rc.y0 = 0;
        rc.y1 = target->params.h;
    }

    int w, h, tw = abs(pl_rect_w(rc)), th = abs(pl_rect_h(rc));
    if (pl_shader_output_size(sh, &w, &h) && (w != tw || h != th)) {
        pl_msg(dp->log, PL_LOG_ERR, "Trying to dispatch a shader with explicit output size "
               "requirements %dx%d%s using a target rect of size %dx%d.",
               w, h, sh->transpose ? " (transposed)" : "", tw, th);
        goto done;
    }

    struct pass_timing *timing = get_timing(dp, sh);
    pl_timer timer = params->timer ? params->timer : timing ? timing->timer : NULL;
    const struct plh_pass_exec x = {
        .pass = &sh->pass, .transpose = sh->transpose, .polar_obj = sh->polar_obj,
        .detect_peak = sh->detect_peak, .peak_state = sh->peak_state,
        .src_tex = sh->src_tex, .on_aux = sh->on_aux, .aux_after = sh->aux_after,
    };
    if (params->blend_params) {
        if (!finish_blended(dp, sh, &x, target, rc, timer, params->blend_params))
            goto done;
    } else {
        const int err = plh_pass_execute(dp->gpu, dp->log, &x, target, rc, timer, &dp->noise);
        if (err) {
            pl_msg(dp->log, PL_LOG_ERR, "Failed launching pass '%s': %s",
                   sh_description(sh), plh_strerror(err));
            goto done;
        }
    }
    if (!params->timer)
        drain_timing(dp, timing);
    ok = true;

done:
    pl_dispatch_abort(dp, params->shader);
    return ok;
}

int plh_launch_errdiff(plh_stream stream, const struct plh_errdiff_args *args);

bool pl_dispatch_compute(pl_dispatch dp, const struct pl_dispatch_compute_params *params)
{
    pl_shader sh = *params->shader;
    bool ok = false;

    if (sh->failed) {
        pl_msg(dp->log, PL_LOG_ERR, "Trying to dispatch a failed shader.");
        goto done;
    }
    if (sh->input != PL_SHADER_SIG_NONE) {
        pl_msg(dp->log, PL_LOG_ERR, "Trying to dispatch shader with incompatible signature!");
        goto done;
    }
    if (!pl_shader_is_compute(sh)) {
        pl_msg(dp->log, PL_LOG_ERR, "Trying to dispatch a non-compute shader using "
               "`pl_dispatch_compute`!");
        goto done;
    }

    struct pass_timing *timing = get_timing(dp, sh);
    pl_timer timer = params->timer ? params->timer : timing ? timing->timer : NULL;
    int err;

    if (sh->kind == PLH_SHADER_ERROR_DIFFUSION) {
        if (timer)
            plh_timer_begin(dp->gpu, timer, 0);
        plh_gpu_stamp(dp->gpu, 0);   // (numbered although no texture records it: gpu_hip.c, fence_here)
        err = plh_launch_errdiff(plh_gpu_stream(dp->gpu), sh->errdiff);
        if (timer)
            plh_timer_end(dp->gpu, timer, 0);
    } else {
        // targetless pass (e.g. sample + peak detection): the rendering area
        // must be given, results leave through side buffers only
        if (!params->width || !params->height) {
            pl_msg(dp->log, PL_LOG_ERR, "Trying to dispatch a targetless compute shader "
                   "that uses vertex attributes, this requires specifying the size of the "
                   "effective rendering area!");
            goto done;
        }
        struct plh_pass *pass = &sh->pass;
        memset(&pass->dst, 0, sizeof(pass->dst)); // every store is out of bounds
        pass->width = params->width;
        pass->height = params->height;
        pass->out_scale[0] = 1.0 / params->width;
        pass->out_scale[1] = 1.0 / params->height;
        pass->base_x = pass->base_y = 0;
        pass->dir_x = pass->dir_y = 1;
        pass->transpose = 0;
        pass->frag_x0 = pass->frag_y0 = 0;
        plh_pass_choose_cells(pass);
        if (!realize_white_noise(dp->gpu, plh_gpu_stream(dp->gpu), &dp->noise, pass))
            goto done;
        if (timer)
            plh_timer_begin(dp->gpu, timer, 0);
        plh_gpu_stamp(dp->gpu, 0);   // (as above)
        err = plh_launch_pass(plh_gpu_stream(dp->gpu), pass);
        if (timer)
            plh_timer_end(dp->gpu, timer, 0);
        if (!err && sh->detect_peak)
            plh_peak_pass_launched(dp->gpu, sh->peak_state, 0, 0, false);
    }

    if (err) {
        pl_msg(dp->log, PL_LOG_ERR, "Failed launching compute shader '%s': %s",
               sh_description(sh), plh_strerror(err));
        goto done;
    }
    if (!params->timer)
        drain_timing(dp, timing);
    ok = true;

done:
    pl_dispatch_abort(dp, params->shader);
    return ok;
}
