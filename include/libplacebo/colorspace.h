/*
 * libplacebo-hip: colour representation / colour space descriptions and the
 * host-side colour maths (Tier-0).
 *
 * API-compatible with the reference's src/include/libplacebo/colorspace.h for
 * the parts the render hot path uses: enums (:29-60,176-240), pl_color_repr
 * (:118-150), pl_hdr_metadata / pl_color_space (:383-520), matrices
 * (:600-680), pl_color_repr_decode (:700), pl_dovi_metadata (:132-149). ICC profiles are
 * carried, not interpreted.
 */
#ifndef LIBPLACEBO_COLORSPACE_H_
#define LIBPLACEBO_COLORSPACE_H_

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#include <libplacebo/common.h>

PL_API_BEGIN

enum pl_color_system {
    PL_COLOR_SYSTEM_UNKNOWN = 0,
    PL_COLOR_SYSTEM_BT_601,
    PL_COLOR_SYSTEM_BT_709,
    PL_COLOR_SYSTEM_SMPTE_240M,
    PL_COLOR_SYSTEM_BT_2020_NC,
    PL_COLOR_SYSTEM_BT_2020_C,
    PL_COLOR_SYSTEM_BT_2100_PQ,
    PL_COLOR_SYSTEM_BT_2100_HLG,
    PL_COLOR_SYSTEM_DOLBYVISION,    // needs pl_color_repr.dovi; input only
    PL_COLOR_SYSTEM_YCGCO,
    PL_COLOR_SYSTEM_YCGCO_RE,
    PL_COLOR_SYSTEM_YCGCO_RO,
    PL_COLOR_SYSTEM_RGB,
    PL_COLOR_SYSTEM_XYZ,
    PL_COLOR_SYSTEM_COUNT
};

PL_API bool pl_color_system_is_ycbcr_like(enum pl_color_system sys);
PL_API bool pl_color_system_is_linear(enum pl_color_system sys);
// (the tables behind the three *_name functions: src/include/libplacebo/colorspace.h:60, :202, :262)
PL_API extern const char *const pl_color_system_names[PL_COLOR_SYSTEM_COUNT];
PL_API const char *pl_color_system_name(enum pl_color_system sys);
PL_API enum pl_color_system pl_color_system_guess_ycbcr(int width, int height);

enum pl_channel {
    PL_CHANNEL_NONE = -1,
    PL_CHANNEL_A = 3,
    PL_CHANNEL_R = 0, PL_CHANNEL_G = 1, PL_CHANNEL_B = 2,
    PL_CHANNEL_Y = 0, PL_CHANNEL_CB = 1, PL_CHANNEL_CR = 2,
    PL_CHANNEL_U = 1, PL_CHANNEL_V = 2,
};

enum pl_color_levels {
    PL_COLOR_LEVELS_UNKNOWN = 0,
    PL_COLOR_LEVELS_LIMITED,
    PL_COLOR_LEVELS_FULL,
    PL_COLOR_LEVELS_COUNT,
    PL_COLOR_LEVELS_TV = PL_COLOR_LEVELS_LIMITED,
    PL_COLOR_LEVELS_PC = PL_COLOR_LEVELS_FULL,
};

enum pl_alpha_mode {
    PL_ALPHA_UNKNOWN = 0,
    PL_ALPHA_INDEPENDENT,
    PL_ALPHA_PREMULTIPLIED,
    PL_ALPHA_NONE,
    PL_ALPHA_MODE_COUNT,
};

struct pl_bit_encoding {
    int sample_depth;   // bits the value is stored / sampled as
    int color_depth;    // bits of actual colour information
    int bit_shift;      // representational left shift
};

PL_API bool pl_bit_encoding_equal(const struct pl_bit_encoding *b1,
                                  const struct pl_bit_encoding *b2);

// Dolby Vision: what an RPU says about a frame (reference colorspace.h:132-149). The decoding
// matrices, and per component a piecewise curve of up to 8 pieces between 9 pivots, each a
// quadratic or a multivariate polynomial (MMR) of all three components.
struct pl_dovi_metadata {
    float nonlinear_offset[3];  // "ycc_to_rgb_offset"
    pl_matrix3x3 nonlinear;     // "ycc_to_rgb", in front of the PQ curve
    pl_matrix3x3 linear;        // "rgb_to_lms", behind it

    struct pl_reshape_data {
        uint8_t num_pivots;
        float pivots[9];        // normalised to [0, 1]
        uint8_t method[8];      // 0 = polynomial, 1 = MMR
        float poly_coeffs[8][3];        // x^0, x^1, x^2 (normalised)
        uint8_t mmr_order[8];           // 1, 2 or 3
        float mmr_constant[8];
        float mmr_coeffs[8][3][7];      // per order
    } comp[3];
};

struct pl_color_repr {
    enum pl_color_system sys;
    enum pl_color_levels levels;
    enum pl_alpha_mode alpha;
    struct pl_bit_encoding bits;
    const struct pl_dovi_metadata *dovi;
};

PL_API extern const struct pl_color_repr pl_color_repr_unknown;
PL_API extern const struct pl_color_repr pl_color_repr_rgb;
PL_API extern const struct pl_color_repr pl_color_repr_sdtv;
PL_API extern const struct pl_color_repr pl_color_repr_hdtv;
PL_API extern const struct pl_color_repr pl_color_repr_uhdtv;
PL_API extern const struct pl_color_repr pl_color_repr_jpeg;

PL_API bool pl_color_repr_equal(const struct pl_color_repr *c1, const struct pl_color_repr *c2);
PL_API void pl_color_repr_merge(struct pl_color_repr *orig, const struct pl_color_repr *update);

// Returns the factor that maps sampled values to the [0,1] range of
// `color_depth`, and normalises `repr->bits` accordingly.
PL_API float pl_color_repr_normalize(struct pl_color_repr *repr);
PL_API enum pl_color_levels pl_color_levels_guess(const struct pl_color_repr *repr);

enum pl_color_primaries {
    PL_COLOR_PRIM_UNKNOWN = 0,
    PL_COLOR_PRIM_BT_601_525,
    PL_COLOR_PRIM_BT_601_625,
    PL_COLOR_PRIM_BT_709,
    PL_COLOR_PRIM_BT_470M,
    PL_COLOR_PRIM_EBU_3213,
    PL_COLOR_PRIM_BT_2020,
    PL_COLOR_PRIM_APPLE,
    PL_COLOR_PRIM_ADOBE,
    PL_COLOR_PRIM_PRO_PHOTO,
    PL_COLOR_PRIM_CIE_1931,
    PL_COLOR_PRIM_DCI_P3,
    PL_COLOR_PRIM_DISPLAY_P3,
    PL_COLOR_PRIM_V_GAMUT,
    PL_COLOR_PRIM_S_GAMUT,
    PL_COLOR_PRIM_FILM_C,
    PL_COLOR_PRIM_ACES_AP0,
    PL_COLOR_PRIM_ACES_AP1,
    PL_COLOR_PRIM_COUNT
};

PL_API bool pl_color_primaries_is_wide_gamut(enum pl_color_primaries prim);
PL_API extern const char *const pl_color_primaries_names[PL_COLOR_PRIM_COUNT];
PL_API const char *pl_color_primaries_name(enum pl_color_primaries prim);
PL_API enum pl_color_primaries pl_color_primaries_guess(int width, int height);

enum pl_color_transfer {
    PL_COLOR_TRC_UNKNOWN = 0,
    PL_COLOR_TRC_BT_1886,
    PL_COLOR_TRC_SRGB,
    PL_COLOR_TRC_LINEAR,
    PL_COLOR_TRC_GAMMA18,
    PL_COLOR_TRC_GAMMA20,
    PL_COLOR_TRC_GAMMA22,
    PL_COLOR_TRC_GAMMA24,
    PL_COLOR_TRC_GAMMA26,
    PL_COLOR_TRC_GAMMA28,
    PL_COLOR_TRC_PRO_PHOTO,
    PL_COLOR_TRC_ST428,
    PL_COLOR_TRC_PQ,
    PL_COLOR_TRC_HLG,
    PL_COLOR_TRC_V_LOG,
    PL_COLOR_TRC_S_LOG1,
    PL_COLOR_TRC_S_LOG2,
    PL_COLOR_TRC_SCRGB,
    PL_COLOR_TRC_COUNT
};

PL_API extern const char *const pl_color_transfer_names[PL_COLOR_TRC_COUNT];
PL_API const char *pl_color_transfer_name(enum pl_color_transfer trc);
PL_API float pl_color_transfer_nominal_peak(enum pl_color_transfer trc);

static inline bool pl_color_transfer_is_hdr(enum pl_color_transfer trc)
{
    return pl_color_transfer_nominal_peak(trc) > 1.0;
}

#define PL_COLOR_SDR_WHITE 203.0f
#define PL_COLOR_SCRGB_WHITE 80.0f
#define PL_COLOR_SDR_CONTRAST 1000.0f
#define PL_COLOR_HDR_BLACK 1e-6f
#define PL_COLOR_HLG_PEAK 1000.0f

struct pl_cie_xy {
    float x, y;
};

static inline struct pl_cie_xy pl_cie_from_XYZ(float X, float Y, float Z)
{
    float k = 1.0f / (X + Y + Z);
    struct pl_cie_xy xy = { k * X, k * Y };
    return xy;
}

static inline float pl_cie_X(struct pl_cie_xy xy) { return xy.x / xy.y; }
static inline float pl_cie_Z(struct pl_cie_xy xy) { return (1 - xy.x - xy.y) / xy.y; }

static inline bool pl_cie_xy_equal(const struct pl_cie_xy *a, const struct pl_cie_xy *b)
{
    return a->x == b->x && a->y == b->y;
}

PL_API struct pl_cie_xy pl_daylight_from_temp(float temperature);
PL_API struct pl_cie_xy pl_blackbody_from_temp(float temperature);
PL_API struct pl_cie_xy pl_white_from_temp(float temperature);

struct pl_raw_primaries {
    struct pl_cie_xy red, green, blue, white;
};

PL_API bool pl_raw_primaries_equal(const struct pl_raw_primaries *a,
                                   const struct pl_raw_primaries *b);
PL_API bool pl_raw_primaries_similar(const struct pl_raw_primaries *a,
                                     const struct pl_raw_primaries *b);
PL_API void pl_raw_primaries_merge(struct pl_raw_primaries *orig,
                                   const struct pl_raw_primaries *update);
PL_API const struct pl_raw_primaries *pl_raw_primaries_get(enum pl_color_primaries prim);

enum pl_hdr_scaling {
    PL_HDR_NORM = 0,    // 1.0 = PL_COLOR_SDR_WHITE
    PL_HDR_SQRT,
    PL_HDR_NITS,
    PL_HDR_PQ,
    PL_HDR_SCALING_COUNT,
};

PL_API float pl_hdr_rescale(enum pl_hdr_scaling from, enum pl_hdr_scaling to, float x);

enum pl_hdr_metadata_type {
    PL_HDR_METADATA_ANY = 0,
    PL_HDR_METADATA_NONE,
    PL_HDR_METADATA_HDR10,
    PL_HDR_METADATA_HDR10PLUS,
    PL_HDR_METADATA_CIE_Y,
    PL_HDR_METADATA_TYPE_COUNT,
};

struct pl_hdr_bezier {
    float target_luma;
    float knee_x, knee_y;
    float anchors[15];
    uint8_t num_anchors;
};

struct pl_hdr_metadata {
    struct pl_raw_primaries prim;
    float min_luma, max_luma;   // cd/m²
    float max_cll;
    float max_fall;
    float scene_max[3];
    float scene_avg;
    struct pl_hdr_bezier ootf;
    float max_pq_y;
    float avg_pq_y;
};

PL_API extern const struct pl_hdr_metadata pl_hdr_metadata_empty;
PL_API extern const struct pl_hdr_metadata pl_hdr_metadata_hdr10;

PL_API bool pl_hdr_metadata_equal(const struct pl_hdr_metadata *a,
                                  const struct pl_hdr_metadata *b);
PL_API void pl_hdr_metadata_merge(struct pl_hdr_metadata *orig,
                                  const struct pl_hdr_metadata *update);
PL_API bool pl_hdr_metadata_contains(const struct pl_hdr_metadata *data,
                                     enum pl_hdr_metadata_type type);

enum pl_rendering_intent {
    PL_INTENT_AUTO = -1,
    PL_INTENT_PERCEPTUAL = 0,
    PL_INTENT_RELATIVE_COLORIMETRIC = 1,
    PL_INTENT_SATURATION = 2,
    PL_INTENT_ABSOLUTE_COLORIMETRIC = 3
};

struct pl_color_space {
    enum pl_color_primaries primaries;
    enum pl_color_transfer transfer;
    struct pl_hdr_metadata hdr;
};

#define pl_color_space(...) (&(struct pl_color_space) { __VA_ARGS__ })

PL_API bool pl_color_space_is_hdr(const struct pl_color_space *csp);
PL_API bool pl_color_space_is_black_scaled(const struct pl_color_space *csp);

// CPU versions of the transfer functions (in place, NORM scaling)
PL_API void pl_color_linearize(const struct pl_color_space *csp, float color[3]);
PL_API void pl_color_delinearize(const struct pl_color_space *csp, float color[3]);

struct pl_nominal_luma_params {
    const struct pl_color_space *color;
    enum pl_hdr_metadata_type metadata;
    enum pl_hdr_scaling scaling;
    float *out_min;
    float *out_max;
    float *out_avg;
};

#define pl_nominal_luma_params(...) (&(struct pl_nominal_luma_params) { __VA_ARGS__ })

PL_API void pl_color_space_nominal_luma_ex(const struct pl_nominal_luma_params *params);

PL_API void pl_color_space_merge(struct pl_color_space *orig, const struct pl_color_space *update);
PL_API bool pl_color_space_equal(const struct pl_color_space *c1, const struct pl_color_space *c2);
PL_API void pl_color_space_infer(struct pl_color_space *space);
PL_API void pl_color_space_infer_ref(struct pl_color_space *space, const struct pl_color_space *ref);
PL_API void pl_color_space_infer_map(struct pl_color_space *src, struct pl_color_space *dst);

PL_API extern const struct pl_color_space pl_color_space_unknown;
PL_API extern const struct pl_color_space pl_color_space_srgb;
PL_API extern const struct pl_color_space pl_color_space_bt709;
PL_API extern const struct pl_color_space pl_color_space_hdr10;
PL_API extern const struct pl_color_space pl_color_space_bt2020_hlg;
PL_API extern const struct pl_color_space pl_color_space_monitor;

struct pl_color_adjustment {
    float brightness;
    float contrast;
    float saturation;
    float hue;
    float gamma;
    float temperature;
};

#define PL_COLOR_ADJUSTMENT_NEUTRAL \
    .contrast       = 1.0,           \
    .saturation     = 1.0,           \
    .gamma          = 1.0,

#define pl_color_adjustment(...) \
    (&(struct pl_color_adjustment) { PL_COLOR_ADJUSTMENT_NEUTRAL __VA_ARGS__ })
PL_API extern const struct pl_color_adjustment pl_color_adjustment_neutral;

enum pl_chroma_location {
    PL_CHROMA_UNKNOWN = 0,
    PL_CHROMA_LEFT,
    PL_CHROMA_CENTER,
    PL_CHROMA_TOP_LEFT,
    PL_CHROMA_TOP_CENTER,
    PL_CHROMA_BOTTOM_LEFT,
    PL_CHROMA_BOTTOM_CENTER,
    PL_CHROMA_COUNT,
};

PL_API void pl_chroma_location_offset(enum pl_chroma_location loc, float *x, float *y);

PL_API pl_matrix3x3 pl_get_rgb2xyz_matrix(const struct pl_raw_primaries *prim);
PL_API pl_matrix3x3 pl_get_xyz2rgb_matrix(const struct pl_raw_primaries *prim);
PL_API pl_matrix3x3 pl_get_color_mapping_matrix(const struct pl_raw_primaries *src,
                                                const struct pl_raw_primaries *dst,
                                                enum pl_rendering_intent intent);
PL_API pl_matrix3x3 pl_get_adaptation_matrix(struct pl_cie_xy src, struct pl_cie_xy dst);
PL_API bool pl_primaries_superset(const struct pl_raw_primaries *a,
                                  const struct pl_raw_primaries *b);
PL_API bool pl_primaries_valid(const struct pl_raw_primaries *prim);
PL_API bool pl_primaries_compatible(const struct pl_raw_primaries *a,
                                    const struct pl_raw_primaries *b);
PL_API struct pl_raw_primaries pl_primaries_clip(const struct pl_raw_primaries *src,
                                                 const struct pl_raw_primaries *dst);

// IPTPQc4 perceptual space used by tone / gamut mapping
PL_API pl_matrix3x3 pl_ipt_rgb2lms(const struct pl_raw_primaries *prim);
PL_API pl_matrix3x3 pl_ipt_lms2rgb(const struct pl_raw_primaries *prim);
// Colour blindness simulation (reference colorspace.h :664-708)
enum pl_cone {
    PL_CONE_L = 1 << 0, PL_CONE_M = 1 << 1, PL_CONE_S = 1 << 2,
    PL_CONE_NONE = 0,
    PL_CONE_LM  = PL_CONE_L | PL_CONE_M,
    PL_CONE_MS  = PL_CONE_M | PL_CONE_S,
    PL_CONE_LS  = PL_CONE_L | PL_CONE_S,
    PL_CONE_LMS = PL_CONE_L | PL_CONE_M | PL_CONE_S,
};

struct pl_cone_params {
    enum pl_cone cones; // cones affected by the vision model
    float strength;     // 1.0 = unaffected, 0.0 = full blindness (> 1 counteracts)
};

#define pl_cone_params(...) (&(struct pl_cone_params) { __VA_ARGS__ })

PL_API extern const struct pl_cone_params pl_vision_normal, pl_vision_protanomaly,
    pl_vision_protanopia, pl_vision_deuteranomaly, pl_vision_deuteranopia, pl_vision_tritanomaly,
    pl_vision_tritanopia, pl_vision_monochromacy, pl_vision_achromatopsia;

// Matrix applying the cone model to linear RGB of the given primaries
PL_API pl_matrix3x3 pl_get_cone_matrix(const struct pl_cone_params *params,
                                       const struct pl_raw_primaries *prim);

PL_API extern const pl_matrix3x3 pl_ipt_lms2ipt;
PL_API extern const pl_matrix3x3 pl_ipt_ipt2lms;

// Affine transform that decodes `repr` into normalised full-range RGB
// (updates `repr` to describe the decoded signal).
PL_API pl_transform3x3 pl_color_repr_decode(struct pl_color_repr *repr,
                                            const struct pl_color_adjustment *params);

// An ICC profile blob (reference colorspace.h:731-758). `data` == NULL means "no profile";
// `signature` must identify the contents. This build carries the description through
// pl_frame but cannot interpret it (no lcms2), see shaders/icc.h.
struct pl_icc_profile {
    const void *data;
    size_t len;
    uint64_t signature;
};

#define pl_icc_profile(...) &(struct pl_icc_profile) { __VA_ARGS__ }

// Compares signatures (and sizes), not contents
PL_API bool pl_icc_profile_equal(const struct pl_icc_profile *p1,
                                 const struct pl_icc_profile *p2);
// signature := digest of the profile bytes (0 without a profile)
PL_API void pl_icc_profile_compute_signature(struct pl_icc_profile *profile);

PL_API_END

#endif // LIBPLACEBO_COLORSPACE_H_
