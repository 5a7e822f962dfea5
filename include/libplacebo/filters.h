/*
 * libplacebo-hip: scaler filter kernels (Tier-0 host maths).
 *
 * API-compatible with the reference's src/include/libplacebo/filters.h
 * (struct layouts at filters.h:39-57, 161-208, 330-410): same type names,
 * same field names, same entry points. The generated LUTs are bit-identical
 * to the reference's src/filters.c (pinned by tests/test_tier0_ref.py against
 * oracle/_ref).
 */
#ifndef LIBPLACEBO_FILTER_KERNELS_H_
#define LIBPLACEBO_FILTER_KERNELS_H_

#include <stdbool.h>
#include <libplacebo/log.h>

PL_API_BEGIN

#define PL_FILTER_MAX_PARAMS 2

// Evaluation context handed to a kernel/window function.
struct pl_filter_ctx {
    float radius;
    float params[PL_FILTER_MAX_PARAMS];
};

// A 1-D weighting function on [0, radius].
struct pl_filter_function {
    const char *name;
    float radius;       // natural radius of the function
    bool resizable;     // radius may be overridden by the config
    bool tunable[PL_FILTER_MAX_PARAMS];
    float params[PL_FILTER_MAX_PARAMS];
    double (*weight)(const struct pl_filter_ctx *f, double x);
    bool opaque;        // cannot be sampled (shader-only, e.g. oversample)
};

PL_API bool pl_filter_function_eq(const struct pl_filter_function *a,
                                  const struct pl_filter_function *b);

PL_API extern const struct pl_filter_function pl_filter_function_box;
PL_API extern const struct pl_filter_function pl_filter_function_triangle;
PL_API extern const struct pl_filter_function pl_filter_function_cosine;
PL_API extern const struct pl_filter_function pl_filter_function_hann;
PL_API extern const struct pl_filter_function pl_filter_function_hamming;
PL_API extern const struct pl_filter_function pl_filter_function_welch;
PL_API extern const struct pl_filter_function pl_filter_function_kaiser;
PL_API extern const struct pl_filter_function pl_filter_function_blackman;
PL_API extern const struct pl_filter_function pl_filter_function_bohman;
PL_API extern const struct pl_filter_function pl_filter_function_gaussian;
PL_API extern const struct pl_filter_function pl_filter_function_quadratic;
PL_API extern const struct pl_filter_function pl_filter_function_sinc;
PL_API extern const struct pl_filter_function pl_filter_function_jinc;
PL_API extern const struct pl_filter_function pl_filter_function_sphinx;
PL_API extern const struct pl_filter_function pl_filter_function_cubic;
PL_API extern const struct pl_filter_function pl_filter_function_hermite;
PL_API extern const struct pl_filter_function pl_filter_function_spline16;
PL_API extern const struct pl_filter_function pl_filter_function_spline36;
PL_API extern const struct pl_filter_function pl_filter_function_spline64;
PL_API extern const struct pl_filter_function pl_filter_function_oversample;
// named members of the cubic family, kept as objects of their own by the reference's older
// configuration API (filters.h:142-147; the configs pl_filter_bicubic ... supersede them)
PL_DEPRECATED_IN(v6.341) PL_API extern const struct pl_filter_function pl_filter_function_bicubic;
PL_DEPRECATED_IN(v6.341) PL_API extern const struct pl_filter_function pl_filter_function_bcspline;
PL_DEPRECATED_IN(v6.341) PL_API extern const struct pl_filter_function pl_filter_function_catmull_rom;
PL_DEPRECATED_IN(v6.341) PL_API extern const struct pl_filter_function pl_filter_function_mitchell;
PL_DEPRECATED_IN(v6.341) PL_API extern const struct pl_filter_function pl_filter_function_robidoux;
PL_DEPRECATED_IN(v6.341) PL_API extern const struct pl_filter_function pl_filter_function_robidouxsharp;

// NULL-terminated list of all functions, and lookup by name.
PL_API extern const struct pl_filter_function * const pl_filter_functions[];
PL_API extern const int pl_num_filter_functions;
PL_API const struct pl_filter_function *pl_find_filter_function(const char *name);

// The older name -> function table (filters.h:175-185): how applications written against the
// previous configuration API select a kernel / window by name. {0}-terminated; "none" -> NULL.
struct pl_filter_function_preset {
    const char *name;
    const struct pl_filter_function *function;
};
PL_API extern const struct pl_filter_function_preset pl_filter_function_presets[];
PL_API extern const int pl_num_filter_function_presets; // excluding the trailing {0}
PL_API const struct pl_filter_function_preset *pl_find_filter_function_preset(const char *name);

enum pl_filter_usage {
    PL_FILTER_UPSCALING    = (1 << 0),
    PL_FILTER_DOWNSCALING  = (1 << 1),
    PL_FILTER_FRAME_MIXING = (1 << 2),

    PL_FILTER_SCALING = PL_FILTER_UPSCALING | PL_FILTER_DOWNSCALING,
    PL_FILTER_ALL     = PL_FILTER_SCALING | PL_FILTER_FRAME_MIXING,
};

// A complete filter: kernel × optional window, plus tuning knobs.
struct pl_filter_config {
    const char *name;
    const char *description;
    enum pl_filter_usage allowed;
    enum pl_filter_usage recommended;

    const struct pl_filter_function *kernel;
    const struct pl_filter_function *window;
    float radius;       // overrides kernel->radius if kernel->resizable
    float params[PL_FILTER_MAX_PARAMS];
    float wparams[PL_FILTER_MAX_PARAMS];
    float clamp;        // 0..1, scales down negative lobes
    float blur;         // >1 blurs, <1 sharpens (0 = 1)
    float taper;        // flat-top width
    bool polar;         // 2-D radial (EWA) instead of separable
    float antiring;     // 0..1
};

PL_API bool pl_filter_config_eq(const struct pl_filter_config *a,
                                const struct pl_filter_config *b);

// Sample the configured filter at offset x (in source texels).
PL_API double pl_filter_sample(const struct pl_filter_config *c, double x);

PL_API extern const struct pl_filter_config pl_filter_spline16;
PL_API extern const struct pl_filter_config pl_filter_spline36;
PL_API extern const struct pl_filter_config pl_filter_spline64;
PL_API extern const struct pl_filter_config pl_filter_nearest;
PL_API extern const struct pl_filter_config pl_filter_box;
PL_API extern const struct pl_filter_config pl_filter_bilinear;
PL_API extern const struct pl_filter_config pl_filter_gaussian;
PL_API extern const struct pl_filter_config pl_filter_sinc;
PL_API extern const struct pl_filter_config pl_filter_lanczos;
PL_API extern const struct pl_filter_config pl_filter_ginseng;
PL_API extern const struct pl_filter_config pl_filter_ewa_jinc;
PL_API extern const struct pl_filter_config pl_filter_ewa_lanczos;
PL_API extern const struct pl_filter_config pl_filter_ewa_lanczossharp;
PL_API extern const struct pl_filter_config pl_filter_ewa_lanczos4sharpest;
PL_API extern const struct pl_filter_config pl_filter_ewa_ginseng;
PL_API extern const struct pl_filter_config pl_filter_ewa_hann;
PL_API extern const struct pl_filter_config pl_filter_bicubic;
PL_API extern const struct pl_filter_config pl_filter_hermite;
PL_API extern const struct pl_filter_config pl_filter_catmull_rom;
PL_API extern const struct pl_filter_config pl_filter_mitchell;
PL_API extern const struct pl_filter_config pl_filter_mitchell_clamp;
PL_API extern const struct pl_filter_config pl_filter_robidoux;
PL_API extern const struct pl_filter_config pl_filter_robidouxsharp;
PL_API extern const struct pl_filter_config pl_filter_ewa_robidoux;
PL_API extern const struct pl_filter_config pl_filter_ewa_robidouxsharp;
PL_API extern const struct pl_filter_config pl_filter_oversample;

#define pl_filter_triangle pl_filter_bilinear

PL_API extern const struct pl_filter_config * const pl_filter_configs[];
PL_API extern const int pl_num_filter_configs;
PL_API const struct pl_filter_config *
pl_find_filter_config(const char *name, enum pl_filter_usage usage);

// The older name -> config table (filters.h:316-329): what mpv / vf_libplacebo style option
// parsers call to select "ewa_lanczos" by name. {0}-terminated; "none" -> NULL filter (built-in
// sampling); aliases have no description.
struct pl_filter_preset {
    const char *name;
    const struct pl_filter_config *filter;
    const char *description;
};
PL_API extern const struct pl_filter_preset pl_filter_presets[];
PL_API extern const int pl_num_filter_presets; // excluding the trailing {0}
PL_API const struct pl_filter_preset *pl_find_filter_preset(const char *name);

struct pl_filter_params {
    struct pl_filter_config config;
    int lut_entries;        // required
    float cutoff;           // truncate the kernel where |w| <= cutoff
    int max_row_size;       // separable only
    int row_stride_align;   // separable only
    float filter_scale;     // deprecated since v6.316, ignored (no effect in the reference either)
};

#define pl_filter_params(...) (&(struct pl_filter_params) { __VA_ARGS__ })

// A sampled filter: 1-D radial LUT (polar) or lut_entries × row_stride
// per-phase weight rows (separable).
typedef const struct pl_filter_t {
    struct pl_filter_params params;
    float radius;           // cut-off radius actually used
    float radius_zero;      // first zero crossing (main lobe)
    const float *weights;
    int row_size;
    bool insufficient;
    int row_stride;
    float radius_cutoff;    // deprecated since v6.336: always equal to `radius`
} *pl_filter;

PL_API pl_filter pl_filter_generate(pl_log log, const struct pl_filter_params *params);
PL_API void pl_filter_free(pl_filter *filter);

// Effective support radius of a config (kernel radius × blur).
// (internal helper of the reference, src/filters.h:22-26; exported here since
// the sampling layer lives in a separate translation unit set.)
PL_API float pl_filter_radius_bound(const struct pl_filter_config *c);

PL_API_END

#endif // LIBPLACEBO_FILTER_KERNELS_H_
