/*
 * libplacebo-hip: high-level renderer, the pl_render_image hot path.
 * API-compatible subset of the reference's src/include/libplacebo/renderer.h
 * (pl_renderer :38-80, pl_render_params :130-380, pl_plane :404-472,
 * pl_frame :528-655, pl_render_image :731).
 *
 * Field names, meanings, defaults AND LAYOUTS are the reference's: `struct pl_frame`,
 * `struct pl_render_params` etc. built against libplacebo's own headers can be passed in
 * unchanged (tests/test_abi_layout.py). Members that select features outside the hot-path
 * scope (hooks, ICC, overlays, film grain, deinterlacing, distortion,
 * blurred borders) are declared at their reference offsets and refused or ignored at run
 * time as documented on each member.
 * Images and targets may be packed, semi-planar or planar / subsampled.
 */
#ifndef LIBPLACEBO_RENDERER_H_
#define LIBPLACEBO_RENDERER_H_

#include <libplacebo/colorspace.h>
#include <libplacebo/dispatch.h>
#include <libplacebo/filters.h>
#include <libplacebo/gpu.h>
#include <libplacebo/shaders/colorspace.h>
#include <libplacebo/shaders/deinterlacing.h>
#include <libplacebo/shaders/dithering.h>
#include <libplacebo/shaders/film_grain.h>
#include <libplacebo/shaders/icc.h>
#include <libplacebo/shaders/lut.h>
#include <libplacebo/shaders/sampling.h>

PL_API_BEGIN

// Thread-safety: Unsafe (one renderer per stream / GPU, like the reference)
typedef struct pl_renderer_t *pl_renderer;

enum pl_render_error {
    PL_RENDER_ERR_NONE              = 0,
    PL_RENDER_ERR_FBO               = 1 << 0,
    PL_RENDER_ERR_SAMPLING          = 1 << 1,
    PL_RENDER_ERR_DEBANDING         = 1 << 2,
    PL_RENDER_ERR_BLENDING          = 1 << 3,
    PL_RENDER_ERR_OVERLAY           = 1 << 4,
    PL_RENDER_ERR_PEAK_DETECT       = 1 << 5,
    PL_RENDER_ERR_FILM_GRAIN        = 1 << 6,
    PL_RENDER_ERR_FRAME_MIXING      = 1 << 7,
    PL_RENDER_ERR_DEINTERLACING     = 1 << 8,
    PL_RENDER_ERR_ERROR_DIFFUSION   = 1 << 9,
    PL_RENDER_ERR_HOOKS             = 1 << 10,
    PL_RENDER_ERR_CONTRAST_RECOVERY = 1 << 11,
    PL_RENDER_ERR_BLUR              = 1 << 12,
};

struct pl_render_errors {
    enum pl_render_error errors;
    const uint64_t *disabled_hooks; // signatures; always NULL here (no hooks)
    int num_disabled_hooks;
};

PL_API pl_renderer pl_renderer_create(pl_log log, pl_gpu gpu);
PL_API void pl_renderer_destroy(pl_renderer *rr);
PL_API struct pl_render_errors pl_renderer_get_errors(pl_renderer rr);
PL_API void pl_renderer_reset_errors(pl_renderer rr, const struct pl_render_errors *errors);

enum pl_clear_mode {
    PL_CLEAR_COLOR = 0, // set to the background colour
    PL_CLEAR_TILES,     // blend against / fill with the two-colour tile pattern
    PL_CLEAR_SKIP,      // leave untouched
    PL_CLEAR_BLUR,      // (unsupported: treated as PL_CLEAR_COLOR)
    PL_CLEAR_MODE_COUNT,
};

enum pl_render_stage {
    PL_RENDER_STAGE_FRAME,
    PL_RENDER_STAGE_BLEND,
    PL_RENDER_STAGE_COUNT,
};

struct pl_render_info {
    const struct pl_dispatch_info *pass;
    enum pl_render_stage stage;
    int index;
    int count;
};

// Where in the pipeline a pl_custom_lut acts (reference renderer.h :83-100)
enum pl_lut_type {
    PL_LUT_UNKNOWN = 0,
    PL_LUT_NATIVE,      // on the raw image contents (after fixing the bit depth)
    PL_LUT_NORMALIZED,  // on normalized (HDR) RGB values
    PL_LUT_CONVERSION,  // replaces the colour conversion (image LUT: native -> RGB; params LUT:
                        // image -> target colour space; target LUT: RGB -> native)
};

struct pl_render_params {
    // Scalers: NULL = built-in bilinear ("free" sampling in the final pass)
    const struct pl_filter_config *upscaler;
    const struct pl_filter_config *downscaler;
    const struct pl_filter_config *plane_upscaler;      // chroma planes; NULL = upscaler
    const struct pl_filter_config *plane_downscaler;    // chroma planes; NULL = downscaler
    float antiringing_strength;
    const struct pl_filter_config *frame_mixer;         // pl_render_image_mix only

    const struct pl_deband_params *deband_params;
    const struct pl_sigmoid_params *sigmoid_params;
    const struct pl_color_adjustment *color_adjustment;
    const struct pl_peak_detect_params *peak_detect_params;
    const struct pl_color_map_params *color_map_params;
    const struct pl_dither_params *dither_params;
    const struct pl_error_diffusion_kernel *error_diffusion;

    const struct pl_cone_params *cone_params; // colour blindness simulation (NULL = off)
    const struct pl_blend_params *blend_params;             // the frame is blended INTO the target
    const struct pl_deinterlace_params *deinterlace_params; // frames with a `field` are deinterlaced
    const struct pl_distort_params *distort_params;         // shaders/sampling.h: an affine map of the image
    const struct pl_hook * const *hooks;                    // unsupported, must be NULL
    int num_hooks;
    const struct pl_custom_lut *lut;    // applied between the image's and the target's colour
    enum pl_lut_type lut_type;          // space, see pl_lut_type

    enum pl_clear_mode background;
    enum pl_clear_mode border;
    float background_color[3];
    float background_transparency;
    float tile_colors[2][3];
    int tile_size;
    float blur_radius;
    float corner_rounding;              // unsupported (ignored)

    bool skip_anti_aliasing;
    bool preserve_mixing_cache;
    bool skip_caching_single_frame;
    bool disable_linear_scaling;
    bool disable_builtin_scalers;
    bool correct_subpixel_offsets;
    bool force_dither;
    bool disable_dither_gamma_correction;
    bool disable_fbos;
    bool force_low_bit_depth_fbos;
    bool dynamic_constants;

    void (*info_callback)(void *priv, const struct pl_render_info *info);
    void *info_priv;

    // Members of older API levels, at the reference's offsets (:350-368)
    bool allow_delayed_peak_detect;         // since v6.254: peak_detect_params->allow_delayed
    const struct pl_icc_params *icc_params; // ignored (no ICC)
    bool ignore_icc_profiles;               // ignored
    int lut_entries;                        // ignored: scaler LUTs have 256 entries
    float polar_cutoff;                     // ignored: 1e-3
    bool skip_target_clearing;              // honoured: = PL_CLEAR_SKIP for background + border
    bool blend_against_tiles;               // honoured: = PL_CLEAR_TILES for the background
};

#define PL_RENDER_DEFAULTS                              \
    .color_map_params   = &pl_color_map_default_params, \
    .color_adjustment   = &pl_color_adjustment_neutral, \
    .tile_colors        = {{0.93, 0.93, 0.93},          \
                           {0.87, 0.87, 0.87}},         \
    .tile_size          = 32,                           \
    .blur_radius        = 16.0,

#define pl_render_params(...) (&(struct pl_render_params) { PL_RENDER_DEFAULTS __VA_ARGS__ })
PL_API extern const struct pl_render_params pl_render_fast_params;
PL_API extern const struct pl_render_params pl_render_default_params;
PL_API extern const struct pl_render_params pl_render_high_quality_params;

#define PL_MAX_PLANES 4

struct pl_plane {
    pl_tex texture;
    enum pl_tex_address_mode address_mode;
    bool flipped;
    int components;           // number of relevant components
    int component_mapping[4]; // semantic index of each component
    float shift_x, shift_y;   // sample position relative to the reference plane's grid
};

enum pl_overlay_mode {
    PL_OVERLAY_NORMAL = 0,
    PL_OVERLAY_MONOCHROME,
    PL_OVERLAY_MODE_COUNT,
};

enum pl_overlay_coords {
    PL_OVERLAY_COORDS_AUTO = 0,
    PL_OVERLAY_COORDS_SRC_FRAME,
    PL_OVERLAY_COORDS_SRC_CROP,
    PL_OVERLAY_COORDS_DST_FRAME,
    PL_OVERLAY_COORDS_DST_CROP,
    PL_OVERLAY_COORDS_COUNT,
};

struct pl_overlay_part {
    pl_rect2df src;
    pl_rect2df dst;
    float color[4];
};

// On-screen display / subtitle bitmaps: a texture and a list of parts of it, each placed on a
// rectangle of the frame (`coords`: of the image or of the target, whole or cropped) and blended
// over it in order. NORMAL: the texture's colour (in `repr` / `color`); MONOCHROME: the part's
// `color`, its alpha times the texture's red channel (glyph atlases).
struct pl_overlay {
    pl_tex tex;
    enum pl_overlay_mode mode;
    enum pl_overlay_coords coords;
    struct pl_color_repr repr;
    struct pl_color_space color;
    const struct pl_overlay_part *parts;
    int num_parts;
};

struct pl_frame {
    int num_planes;           // 1..4 (packed, semi-planar, planar)
    struct pl_plane planes[PL_MAX_PLANES];

    // Interlacing description, filled in by pl_queue (utils/frame_queue.h): the field to show,
    // which field comes first in time, and the neighbouring frames the temporal deinterlacers
    // read (same plane layout and sizes). Deinterlaced with pl_render_params.deinterlace_params,
    // shown woven without.
    enum pl_field field;
    enum pl_field first_field;
    const struct pl_frame *prev, *next;

    bool (*acquire)(pl_gpu gpu, struct pl_frame *frame);
    void (*release)(pl_gpu gpu, struct pl_frame *frame);

    struct pl_color_repr repr;
    struct pl_color_space color;

    pl_icc_object icc;              // ignored (no lcms2 in this build), warned about once
    struct pl_icc_profile profile;  // ignored likewise

    // Optional LUT attached to the frame (images: applied while decoding; targets: while
    // encoding). lut_type 0 = guess from the LUT's repr_in / repr_out.
    const struct pl_custom_lut *lut;
    enum pl_lut_type lut_type;

    pl_rect2df crop;          // 0 = whole frame; flipped rects flip the image
    pl_rotation rotation;     // clockwise, in multiples of 90 degrees (common.h)
    float pixel_aspect_ratio; // informational (0 = square); the renderer never reads it

    const struct pl_overlay *overlays;      // drawn over the frame (images: in mixing, per frame)
    int num_overlays;

    struct pl_film_grain_data film_grain;   // not synthesised, see shaders/film_grain.h

    void *user_data;
};

// Set plane shifts from a chroma sample location (applies to subsampled planes)
PL_API void pl_frame_set_chroma_location(struct pl_frame *frame,
                                         enum pl_chroma_location chroma_loc);

// true if the frame's crop does not cover its whole reference plane
PL_API bool pl_frame_is_cropped(const struct pl_frame *frame);

// Fill every plane of `frame` with one colour, given as sRGB and converted to the frame's colour
// space and encoding (renderer.h:672-690 in the reference; premultiplied frames get rgb * alpha).
// What applications call on a target before / instead of rendering into a part of it.
PL_API void pl_frame_clear_rgba(pl_gpu gpu, const struct pl_frame *frame, const float rgba[4]);

static inline void pl_frame_clear(pl_gpu gpu, const struct pl_frame *frame, const float rgb[3])
{
    const float rgba[4] = { rgb[0], rgb[1], rgb[2], 1.0f };
    pl_frame_clear_rgba(gpu, frame, rgba);
}

// Fill every plane of `frame` with two-colour tiles of `tile_size` reference-plane texels
// (subsampled planes: scaled by their subsampling ratio); alpha is set to 1.
PL_API void pl_frame_clear_tiles(pl_gpu gpu, const struct pl_frame *frame,
                                 const float tile_colors[2][3], int tile_size);

// Fill in what pl_render_image would infer (crop, bit depths, colour spaces)
PL_API void pl_frames_infer(pl_renderer rr, struct pl_frame *image, struct pl_frame *target);

// Render `image` to `target` (`image` == NULL: clear the target to the background colour).
// Returns false on hard failure; soft failures
// disable the offending stage and are reported by pl_renderer_get_errors.
PL_API bool pl_render_image(pl_renderer rr, const struct pl_frame *image,
                            const struct pl_frame *target,
                            const struct pl_render_params *params);

/* ---- frame mixing (reference renderer.h :754-854) ---- */

// A set of frames around the vsync being drawn. `timestamps` are relative to that vsync
// (it is shown at 0.0 and held for `vsync_duration`), in units of one nominal source frame
// duration, sorted ascending; zero-order-hold semantics. `signatures` identify frames across
// calls (rendered frames are cached by signature).
struct pl_frame_mix {
    int num_frames;
    const struct pl_frame **frames;
    const uint64_t *signatures;
    const float *timestamps;
    float vsync_duration;
};

// The radius of frames a mixer needs around the vsync (0 = built-in oversampling: just the
// current and the next frame)
static inline float pl_frame_mix_radius(const struct pl_render_params *params)
{
    if (!params->frame_mixer || !params->frame_mixer->kernel)
        return 0.0;
    return params->frame_mixer->kernel->radius;
}

// Frame shown at the current vsync by zero-order hold / nearest-neighbour semantics, or NULL
PL_API const struct pl_frame *pl_frame_mix_current(const struct pl_frame_mix *mix);
PL_API const struct pl_frame *pl_frame_mix_nearest(const struct pl_frame_mix *mix);

// Generalisation of pl_render_image: every frame with a non-negligible weight under
// `params->frame_mixer` (a 1-D pl_filter_config over time, or oversampling) is rendered -- or
// taken from the cache -- at the output size in the target's colour space, the frames are
// blended in linear light and the result goes through the output stage. Without a frame mixer
// (or with a single frame) the nearest frame is rendered. An empty mix clears the target.
PL_API bool pl_render_image_mix(pl_renderer rr, const struct pl_frame_mix *images,
                                const struct pl_frame *target,
                                const struct pl_render_params *params);

// pl_frames_infer for a mix: adjusts `target` and returns the adjusted reference frame
PL_API void pl_frames_infer_mix(pl_renderer rr, const struct pl_frame_mix *mix,
                                struct pl_frame *target, struct pl_frame *out_ref);

// Drop cached FBOs / mixing frames / LUT state
PL_API void pl_renderer_flush_cache(pl_renderer rr);

// Pre-v6 leftovers that callers still reference (src/include/libplacebo/renderer.h:855-880):
// the {0}-terminated preset lists option parsers walk, and the save / load front ends of the
// gpu's pl_cache (pl_gpu_set_cache).
PL_API extern const struct pl_filter_preset pl_frame_mixers[];
PL_API extern const int pl_num_frame_mixers;    // excluding the trailing {0}
PL_API extern const struct pl_filter_preset pl_scale_filters[];
PL_API extern const int pl_num_scale_filters;   // excluding the trailing {0}
PL_API size_t pl_renderer_save(pl_renderer rr, uint8_t *out_cache);
PL_API void pl_renderer_load(pl_renderer rr, const uint8_t *cache);

// HDR metadata measured by the last frame's peak detection, if any
PL_API bool pl_renderer_get_hdr_metadata(pl_renderer rr, struct pl_hdr_metadata *metadata);

// HIP extension (multi-GPU): the renderer's tone-mapping state object, whose
// pending peak measurement can be all-reduced across ranks through pl_hip_peak_buffer().
PL_API pl_shader_obj pl_hip_renderer_tone_map_state(pl_renderer rr);

PL_API_END

#endif // LIBPLACEBO_RENDERER_H_
