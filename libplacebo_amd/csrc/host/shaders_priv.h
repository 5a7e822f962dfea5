/*
 * libplacebo-hip — internals of the op-recording pl_shader.
 * Counterpart of the reference's src/shaders.h (struct pl_shader_t :85-119,
 * pl_shader_obj_t :147-160, SH_OBJ :175-180).
 */
#ifndef PLH_SHADERS_PRIV_H_
#define PLH_SHADERS_PRIV_H_

#include <libplacebo/shaders.h>
#include <libplacebo/shaders/colorspace.h>
#include <libplacebo/colorspace.h>
#include <libplacebo/dispatch.h>

#include "gpu_priv.h"

enum pl_shader_obj_type {
    PL_SHADER_OBJ_INVALID = 0,
    PL_SHADER_OBJ_COLOR_MAP,
    PL_SHADER_OBJ_SAMPLER,
    PL_SHADER_OBJ_DITHER,
    PL_SHADER_OBJ_LUT,
};

struct pl_shader_obj_t {
    int refcount;
    enum pl_shader_obj_type type;
    pl_gpu gpu;
    void (*uninit)(pl_gpu gpu, void *priv);
    void *priv;
};

enum plh_shader_kind {
    PLH_SHADER_PASS = 0,        // sampler + colour ops -> plh_launch_pass
    PLH_SHADER_ERROR_DIFFUSION, // standalone compute (pl_dispatch_compute)
};

struct plh_errdiff_args;

struct pl_shader_t {
    pl_log log;
    struct pl_shader_params params;
    bool failed;
    bool mutable_;
    enum pl_shader_sig input, output;
    int output_w, output_h;
    bool transpose;
    bool is_compute;
    int group_size[2];
    size_t shmem;

    enum plh_shader_kind kind;
    struct plh_pass pass;
    // persistent objects whose device memory the recorded pass points at
    pl_shader_obj held[16];
    int num_held;
    // the pass carries a brightness measurement for `peak_state` (shader_color.c); the
    // dispatch reports its launch through plh_peak_pass_launched
    bool detect_peak;
    pl_shader_obj peak_state;

    char steps[12][96];         // one name per recorded stage (sh_describef)
    int num_steps;
    pl_shader_info info;        // built by pl_shader_finalize
    char *listing;
    size_t listing_len, listing_cap;
    struct pl_shader_res res;

    // pl_shader_finalize publishes the recorded pass under this ticket (the last line of
    // pl_shader_res.glsl): what pl_pass_create resolves back to the op list (gpu.c)
    uint64_t ticket;

    struct plh_errdiff_args *errdiff;
    // polar sampler state, for the launch-time phase-class setup (shader_sampling.c)
    void *polar_obj;
    // texture and rect bound by the sampling stage (sh_bind), for pass fusion
    pl_tex src_tex;
    pl_rect2df src_rect;
    // run on the measurement stream (set by the renderer on a pass plh_shader_aux_eligible accepts)
    bool on_aux;
    uint64_t aux_after;
    // a slot of the gpu's scratch ring this shader's ops point at (the Dolby Vision reshaping curves,
    // shader_color.c), given back when the shader is dispatched, reset or freed
    const void *scratch;
    pl_gpu scratch_gpu;
};

// true if the only device memory the recorded pass reads is `src_tex` and the measurement's own
// buffers: what the two-stream ordering of gpu_hip.c covers
bool plh_shader_aux_eligible(const pl_shader sh);

struct pl_sample_src;
struct pl_sample_filter_params;

// pl_shader_sample_polar with `pre` (a recorded 1:1 on-grid fetch of a whole texture followed
// by colour ops: the reference's PASS A) folded in as per-source-texel pre-ops: the FBO round
// trip disappears, the numerics (ops, then rgba16hf rounding) stay. Returns false without
// touching `sh` if `pre` cannot be fused. See renderer.c pass_scale_main.
bool plh_shader_sample_polar_fused(pl_shader sh, const pl_shader pre,
                                   const struct pl_sample_src *src,
                                   const struct pl_sample_filter_params *params);

// called by the dispatch once the target geometry of a POLAR pass is known
// What a colour-mapping request resolves to (colormap_plan.c): a pure function of the request.
struct plh_colormap_plan {
    struct pl_color_space src, dst;     // after pl_color_space_infer_map
    bool identity;                      // equal spaces: nothing to map (the rest is unset)
    struct pl_tone_map_params tone;     // end points snapped, function degraded if stateless
    struct pl_gamut_map_params gamut;   // target gamut clipped to the source unless expanding
    bool closed_form;                   // closed-form steps allowed (no state, or LUT not forced)
    bool need_tone, need_gamut;         // steps that remain (need_gamut false when folded)
    bool tone_direct;                   // clip / linear evaluated without a LUT
    bool fold_saturation;               // `saturation` gamut step folded into the output matrix
};
void plh_colormap_resolve(struct plh_colormap_plan *plan, const struct pl_color_map_params *params,
                                 const struct pl_color_space *src, const struct pl_color_space *dst,
                                 bool stateful);

void plh_polar_pp_setup(pl_gpu gpu, pl_log log, void *polar_obj, struct plh_pass *pass);

// (`recorded`: the launch carried the `written` event itself, plh_peak_written_event)
void plh_peak_pass_launched(pl_gpu gpu, pl_shader_obj state, int on, uint64_t seq, bool recorded);
plh_event plh_peak_written_event(pl_shader_obj state);

// the finalized, still-alive shader behind a "#pl_hip_pass <ticket>" line; NULL if there is none
pl_shader plh_shader_from_glsl(const char *glsl);

// What a launch needs beyond the op list, and the launch itself: shared by pl_dispatch_finish
// and pl_pass_run. `noise` is the caller's white-noise plane (dispatch.c: realize_white_noise).
struct plh_pass_exec {
    struct plh_pass *pass;      // modified: target half, polar tables, cell phases
    bool transpose;
    void *polar_obj;
    bool detect_peak;
    pl_shader_obj peak_state;
    int out_w, out_h;           // the shader's own output size requirement, 0 = none
    // two-stream mode (pl_hip_params.async_measure): the texture the sampler reads, and whether
    // the caller wants this pass on the measurement stream (renderer.c: plh_work_texture)
    pl_tex src_tex;
    bool on_aux;
    uint64_t aux_after;         // main-stream stamp of the last upload into the measurement's tables
};
// returns 0, a negative plh error code, or PLH_EXEC_BAD_* (message already logged)
int plh_pass_execute(pl_gpu gpu, pl_log log, const struct plh_pass_exec *x, pl_tex target,
                     pl_rect2d rc, pl_timer timer, pl_buf *noise);

// An overlay (or a rendered pass being blended) as the dispatch takes it: `parts` in target pixels,
// drawn in order. The shader holds colour ops only; for PLH_OVERLAY_MONOCHROME the texture's
// coverage multiplies the colour in front of op `coverage_at` (the plane's swizzle).
struct plh_overlay_draw {
    pl_tex tex;
    int mode;                               // enum plh_overlay_mode
    bool linear;
    bool premultiplied;
    int coverage_at;
    const struct pl_blend_params *blend;    // NULL: the colour replaces the target's
    const struct plh_overlay_part *parts;
    int num_parts;
};
bool plh_dispatch_overlay(pl_dispatch dp, pl_shader *sh, pl_tex target,
                          const struct plh_overlay_draw *draw);
// the two recorded passes of a separable one-component downscale as one launch (dispatch.c):
// 1 = done, 0 = declined (nothing consumed), -1 = failed
int plh_dispatch_lowpass2(pl_dispatch dp, pl_shader *vert, pl_shader *horiz, pl_tex target);

#define SH_GPU(sh) ((sh)->params.gpu)

#define SH_FAIL(sh, ...) do {                       \
        (sh)->failed = true;                        \
        pl_msg((sh)->log, PL_LOG_ERR, __VA_ARGS__); \
    } while (0)

static inline struct pl_glsl_version sh_glsl(const pl_shader sh)
{
    return SH_GPU(sh) ? SH_GPU(sh)->glsl : sh->params.glsl;
}

bool sh_require(pl_shader sh, enum pl_shader_sig insig, int w, int h);
bool sh_try_compute(pl_shader sh, int bw, int bh, bool flex, size_t mem);
void sh_describef(pl_shader sh, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
const char *sh_description(pl_shader sh);    // name of the last recorded stage
void sh_listf(pl_shader sh, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// append a colour op; returns NULL (and fails the shader) if the chain is full
struct plh_op *sh_op(pl_shader sh, int kind);

void *sh_require_obj(pl_shader sh, pl_shader_obj *ptr, enum pl_shader_obj_type type,
                     size_t priv_size, void (*uninit)(pl_gpu gpu, void *priv));
#define SH_OBJ(sh, ptr, type, t, uninit) \
    ((t *) sh_require_obj(sh, ptr, type, sizeof(t), uninit))

// keep `obj` alive until the shader has been dispatched / reset
void sh_hold(pl_shader sh, pl_shader_obj obj);

// Bind a texture for sampling: fills src view + tex_coord corners + pt
// (sh_bind, src/shaders.c:513-571)
bool sh_bind(pl_shader sh, pl_tex tex, enum pl_tex_address_mode address_mode,
             const pl_rect2df *rect);

// The value the reference would embed for a constant printed with "%f"
// (6 decimals), e.g. the PQ constants in shaders/colorspace.c
float plh_fmtf(double v);

#endif // PLH_SHADERS_PRIV_H_
