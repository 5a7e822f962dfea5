/*
 * libplacebo-hip — internals shared by the two halves of the renderer's executor:
 * renderer.c (one frame) and render_mix.c (pl_render_image_mix).
 */
#ifndef PLH_RENDERER_PRIV_H_
#define PLH_RENDERER_PRIV_H_

#include <libplacebo/renderer.h>

#include "render_plan.h"
#include "shaders_priv.h"

#define RR_MAX_FBOS 16
#define RR_MEASURE_FBOS 3
#define RR_MAX_MIX_FRAMES 16        // frames one mix may blend (reference renderer.c:3610)
#define RR_MAX_CACHED_FRAMES 32

// persistent filter state of one scaler slot, per direction
struct scaler_slot {
    pl_shader_obj up, down;
};

enum { RR_LUT_IMAGE, RR_LUT_TARGET, RR_LUT_PARAMS, RR_LUT_COUNT };

// a frame rendered for mixing, kept across vsyncs
struct mix_entry {
    uint64_t signature;
    uint64_t params_digest;     // of the render params it was produced with
    struct pl_color_space color;
    struct pl_color_repr repr;
    pl_rect2df crop;
    pl_tex tex;
    int comps;
    bool stale;                 // not referenced by the mix being built
};

struct pl_renderer_t {
    pl_gpu gpu;
    pl_dispatch dp;
    pl_log log;
    enum pl_render_error errors;

    pl_tex fbos[RR_MAX_FBOS];
    int num_fbos;
    // pl_hip_params.async_measure: the measuring pass of a frame runs on its own stream, beside
    // the previous frame's later passes -- which still read the previous frame's intermediate, so
    // that pass cycles through textures of its own instead of drawing from `fbos` (three: the
    // host runs a frame ahead of the main stream, so the texture of two frames ago may still be read)
    pl_tex measure_fbo[RR_MEASURE_FBOS];
    unsigned measure_flip;

    struct scaler_slot scale_main, scale_ref, scale_contrast;
    struct scaler_slot scale_plane[PL_MAX_PLANES];   // chroma / alpha planes of the image
    struct scaler_slot scale_out[PL_MAX_PLANES];     // planes of a planar target
    pl_shader_obj tone_map_state;
    pl_shader_obj dither_state;
    pl_shader_obj lut_state[RR_LUT_COUNT];
    int last_dither_depth;
    bool warned_icc, warned_grain;
    struct plh_overlay_part *osd_parts;     // the overlay being drawn, placed on the plane
    int osd_cap;

    struct mix_entry cache[RR_MAX_CACHED_FRAMES];
    int num_cached;
    pl_tex spare[RR_MAX_CACHED_FRAMES];     // textures of evicted entries, for reuse
    int num_spare;
};

// The image as it travels through a frame: either recorded-but-not-run (`rec`) or resident
// in a texture (`tex`), never both.
struct work_image {
    pl_shader rec;
    pl_tex tex;
    pl_tex copy_of;             // `rec` is so far nothing but a 1:1 fetch of this texture
    int w, h;
    pl_rect2df rect;
    struct pl_color_repr repr;
    struct pl_color_space color;
    int comps;
    pl_fmt store_as;            // format override for the next flush
    // if the next flush fails: message, error bit to raise, texture to continue with
    const char *fail_msg;
    enum pl_render_error fail_bit;
    pl_tex fail_tex;
};

// everything one pl_render_image call needs
struct frame_job {
    pl_renderer rr;
    const struct pl_render_params *params;
    struct pl_frame image, target;
    struct rp_caps caps;
    struct rp_geometry geo;
    struct work_image img;
    bool fbo_busy[RR_MAX_FBOS];
    bool peak_pending;          // a same-frame measurement rides on `img.rec`
    pl_tex measure_fbo;         // the member of rr->measure_fbo this frame wrote, if any
    pl_tex features_full;       // full-size feature plane written by the measuring pass (renderer.c:
                                // measure_peak), for make_feature_map to start from
    pl_tex features_src;        // ... and the image (resident texture) it was extracted from
    struct pl_color_space features_color;   // ... as this colour space (the detected HDR metadata that
                                // arrives afterwards may change how a non-linear image is linearised)
    bool image_acquired, target_acquired;
    struct pl_frame prev, next;     // deinterlacing: local copies of image.prev / .next while acquired
    bool prev_acquired, next_acquired;
    bool target_borrowed;       // the target belongs to an enclosing job: neither acquire nor release
    struct pl_render_info info;
};

#define RR_LOG(rr, lev, ...) pl_msg((rr)->log, lev, __VA_ARGS__)

// renderer.c
bool plh_job_begin(struct frame_job *job, bool acquire_image);
void plh_job_end(struct frame_job *job);
bool plh_params_supported(pl_renderer rr, const struct pl_render_params *params);
void plh_job_watch_passes(struct frame_job *job);    // route pass timings to info_callback
bool plh_stage_read(struct frame_job *job);
bool plh_stage_scale(struct frame_job *job);
void plh_stage_colors(struct frame_job *job);
bool plh_stage_output(struct frame_job *job);
pl_shader plh_work_shader(struct frame_job *job, struct work_image *img);
pl_tex plh_work_texture(struct frame_job *job, struct work_image *img);
// append a plain 1:1 fetch of another texture to `sh` as a colour op
bool plh_append_plane_fetch(pl_shader sh, const pl_shader fetch, const struct pl_plane *plane);
struct plh_op *plh_append_scale(pl_shader sh, float k, bool with_alpha);

// render_overlay.c (reference draw_overlays, src/renderer.c:811-1020): `overlays` over `fbo`, which
// holds `comps` components (`comp_map`: the plane's) of a frame in `color` / `repr`;
// `output_shift`: target pixels -> texels of `fbo`
void plh_draw_overlays(struct frame_job *job, pl_tex fbo, int comps, const int comp_map[4],
                       const struct pl_overlay *overlays, int num, bool have_image,
                       struct pl_color_space color, struct pl_color_repr repr,
                       const pl_transform2x2 *output_shift);
pl_transform2x2 plh_plane_shift(const struct pl_plane *plane, pl_tex ref);

#endif // PLH_RENDERER_PRIV_H_
