/*
 * libplacebo-hip — the renderer's planner.
 *
 * pl_render_image is split in two halves here. This half is PURE: it looks at frame
 * descriptions (sizes, formats, colour metadata) and pl_render_params and decides what the
 * frame needs -- which rect of which plane is read, which scaler runs in which light, where the
 * HDR peak is measured, what the output stage does -- without touching the GPU or recording a
 * shader. renderer.c is the other half: it executes those decisions. The decisions are the
 * reference's (src/renderer.c, cited per function in render_plan.c); the decomposition is this
 * implementation's, chosen so that the plan of a frame can be printed and asserted on without
 * a GPU (tests/test_render_plan.py).
 */
#ifndef PLH_RENDER_PLAN_H_
#define PLH_RENDER_PLAN_H_

#include <libplacebo/renderer.h>

/* ---- capabilities the planner needs to know about the backend ---- */
struct rp_caps {
    pl_fmt fbo[5];              // intermediate format per component count (fbo[4] == NULL: none)
    size_t max_shmem;           // LDS per workgroup
    bool sampling_broken;       // PL_RENDER_ERR_SAMPLING is set
    bool peak_broken;           // PL_RENDER_ERR_PEAK_DETECT
    bool deband_broken;
    bool contrast_broken;
    bool errdiff_broken;
};

/* ---- geometry ---- */
struct rp_geometry {
    pl_rect2df src;             // image crop, normalised, adjusted for the rounded target rect
    pl_rect2df dstf;            // the rounded target rect (flips live here)
    pl_rect2d dst;
    pl_rotation rotation;       // image rotation relative to the target, [0, 4)
};

// Fit `image_crop` (in texels of an iw x ih reference plane) onto `target_crop` (tw x th).
struct rp_geometry rp_fit_rects(pl_rect2df image_crop, int iw, int ih, pl_rotation image_rot,
                                pl_rect2df target_crop, int tw, int th, pl_rotation target_rot);
// the same without an image: the rounded target rect only
struct rp_geometry rp_fit_target(pl_rect2df target_crop, int tw, int th, pl_rotation target_rot);

/* ---- frames and planes ---- */
enum rp_plane_role {
    RP_PLANE_UNUSED = 0,
    RP_PLANE_ALPHA,
    RP_PLANE_CHROMA,
    RP_PLANE_LUMA,
    RP_PLANE_RGB,
    RP_PLANE_XYZ,
};

enum rp_plane_role rp_plane_role(const struct pl_plane *plane, const struct pl_color_repr *repr);
// index of the plane that defines the frame's pixel grid
int rp_reference_plane(const struct pl_frame *frame);
// fill in what can be derived from the frame itself (primaries guess, alpha mode, bit depth)
void rp_complete_frame(struct pl_frame *frame);
// both frames of a render, in the reference's order (colour spaces are inferred as a pair)
void rp_complete_frames(struct pl_frame *image, struct pl_frame *target);
// NULL if usable, else what is wrong with it
const char *rp_frame_problem(const struct pl_frame *frame, bool is_target);
// which way a custom LUT attached to a frame acts (`reversed`: target side)
enum pl_lut_type rp_frame_lut_type(const struct pl_frame *frame, bool reversed);

struct rp_plane_layout {
    enum rp_plane_role role;
    struct pl_plane plane;      // copy; alpha mapping cleared when the alpha mode is NONE
    pl_rect2df rect;            // the image crop in this plane's texels
    float logical_w, logical_h; // size the plane would have at its (rounded) subsampling ratio
    float neutral[3];           // debanding: value noise is centred on, per sampled component
};

struct rp_image_layout {
    int ref;                                // index of the reference plane
    struct rp_plane_layout planes[PL_MAX_PLANES];
    float neutral_luma, neutral_chroma;     // initial colour before the planes are merged
    pl_rect2d grid;                         // the reference rect, snapped to whole texels
    float off_x, off_y;                     // sub-texel remainder of that snapping
    float stretch_x, stretch_y;             // grid size / exact size
};

void rp_layout_image(const struct pl_frame *image, struct rp_image_layout *out);
// sampling request that brings plane `i` onto the reference grid
struct pl_sample_src rp_plane_request(const struct rp_image_layout *lay, int i);
// true if that request is the identity (the plane already is the grid)
bool rp_plane_request_is_identity(const struct pl_sample_src *req);

/* ---- scalers ---- */
enum rp_scaler_kind {
    RP_SCALER_BUILTIN = 0,  // whatever the texture unit does: bilinear if it can, else nearest
    RP_SCALER_NEAREST,
    RP_SCALER_BICUBIC,
    RP_SCALER_HERMITE,
    RP_SCALER_GAUSSIAN,
    RP_SCALER_OVERSAMPLE,
    RP_SCALER_FILTER,       // pl_filter_config: polar (EWA) or separable
};

enum rp_direction { RP_DIR_NONE = 0, RP_DIR_UP, RP_DIR_DOWN };
enum rp_usage { RP_USE_MAIN, RP_USE_PLANE, RP_USE_LOWPASS };

struct rp_scaler {
    enum rp_scaler_kind kind;
    enum rp_direction dir;          // overall: downscaling wins
    enum rp_direction axis[2];      // x, y
    const struct pl_filter_config *filter;
};

struct rp_scaler rp_pick_scaler(const struct rp_caps *caps, const struct pl_render_params *params,
                                enum rp_usage usage, const struct pl_sample_src *req,
                                pl_fmt src_format);

/* ---- main scaling stage ---- */
struct rp_scale_stage {
    struct rp_scaler scaler;
    bool skip;              // nothing to do (1:1 and resizable)
    bool defer;             // "free" scaling: the next pass samples at the output size
    bool restore_transfer;  // image arrived in linear light but stays non-linear for scaling
    bool linear;            // scale in linear light
    bool sigmoid;           // ... and sigmoidized
    bool peak_before;       // measure the HDR peak before (else after) the scaler
    int out_w, out_h;
};

struct rp_scale_stage rp_plan_scale(const struct rp_caps *caps, const struct pl_render_params *params,
                                    const struct pl_sample_src *req, pl_fmt src_format,
                                    const struct pl_color_space *img_color, int comps,
                                    bool fixed_size_input);

/* ---- HDR peak measurement ---- */
// NULL if the peak should be measured now, else the reason it is not
const char *rp_peak_skip_reason(const struct rp_caps *caps, const struct pl_render_params *params,
                                const struct pl_color_space *image, const struct pl_color_space *img,
                                const struct pl_color_space *target);

/* ---- fusing the pending image into a polar main scaler ---- */
// `pending_lite`: the ops recorded on the pending image need no transcendental (decode, plane map,
// affine ...); `force`: PL_HIP_NO_FUSION as read by the caller, -1 unset, 0 = fuse wherever it is
// valid, 1 = never. An upscale keeps the reference's two passes when its pending ops are not lite
// (LINEARIZE + SIGMOIDIZE: pl_render_default_params on SDR video) -- the intermediate is the small
// side, and fused the polar kernel would stage its tile through the full op interpreter
// (1080p -> 4K: 0.113 ms fused, 0.063 ms as two passes); a downscale keeps the fusion (its
// intermediate is the large side, k_polar_mxd linearises while it stages).
bool rp_fuse_into_polar(enum rp_direction dir, bool pending_lite, float antiring, int force);

/* ---- contrast recovery ---- */
bool rp_wants_feature_map(const struct rp_caps *caps, const struct pl_render_params *params,
                          const struct pl_color_space *img, const struct pl_color_space *target,
                          int out_w, int out_h, int *map_w, int *map_h);

/* ---- output ---- */
enum rp_dither { RP_DITHER_NONE = 0, RP_DITHER_ORDERED, RP_DITHER_ERROR_DIFFUSION };

struct rp_output_plane {
    pl_rect2df exact;       // target rect in this plane's texels (normalised)
    pl_rect2d covered;      // whole texels touched
    pl_rect2d store;        // rect handed to the dispatch (flips applied)
    struct pl_sample_src request;   // planar targets: how the plane samples the finished image
    float ratio_x, ratio_y;
};

struct rp_output_stage {
    enum pl_clear_mode background, border;
    bool premultiply;       // image has alpha that must be premultiplied for blending
    bool blend;             // ... and blended against the background colour
    bool drop_alpha;        // image becomes opaque afterwards
    bool unpremultiply;     // target wants independent alpha
    bool encode;            // pl_shader_encode_color (false: a CONVERSION LUT did it)
    bool delinearize_xyz;
    enum pl_lut_type target_lut;
    struct pl_color_repr repr;  // target repr as handed to the encoder (normalised)
    float scale;                // of the target's integer encoding
    bool transposed;            // odd quarter-turn rotation
    pl_rect2d dst;              // target rect in the orientation the stores use
    bool clear_border;
    int dither_depth;           // 0 = no dithering of any kind
    int num_planes;
    struct rp_output_plane planes[PL_MAX_PLANES];
};

void rp_plan_output(const struct pl_render_params *params, const struct pl_frame *target,
                    const struct rp_geometry *geo, int img_comps, enum pl_alpha_mode img_alpha,
                    struct rp_output_stage *out);
enum rp_dither rp_pick_dither(const struct rp_caps *caps, const struct pl_render_params *params,
                              int depth, int plane_h);

/* ---- human-readable plan (tests, PL_LOG_DEBUG) ---- */
struct rp_summary {
    char text[2048];
};

// Plan one frame from descriptions alone and print the decisions, one per line.
void rp_summarise(const struct rp_caps *caps, const struct pl_frame *image,
                  const struct pl_frame *target, const struct pl_render_params *params,
                  struct rp_summary *out);

#endif // PLH_RENDER_PLAN_H_
