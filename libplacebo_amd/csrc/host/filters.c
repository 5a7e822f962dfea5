/*
 * libplacebo-hip — Tier-0 host maths: scaler filter kernels and LUT generation.
 *
 * Fresh implementation of the behaviour of the reference's src/filters.c:
 *   pl_filter_sample      (filters.c:82-124)   kernel(x/blur, taper) × window
 *   cutoff root search    (filters.c:126-151)  0.01-step scan + secant refine
 *   separable rows        (filters.c:155-177)  per-phase rows normalised to Σ=1
 *   pl_filter_generate    (filters.c:186-245)  polar 1-D LUT / separable 2-D LUT
 *   function & config tables (filters.c:254-976)
 *
 * Every expression that feeds a LUT entry is evaluated in the same precision
 * and operation order as the reference so the float LUTs come out
 * bit-identical (checked against oracle/_ref by tests/test_tier0_ref.py).
 * Runs once per filter/ratio change (256 evaluations); never per pixel.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

// (this file DEFINES the objects the header marks as deprecated)
#define PL_DEPRECATED_IN(VER)
#include <libplacebo/filters.h>
#include "host_common.h"

/* ------------------------------------------------------------------------ */
/* weighting functions: double in, double out, x already folded to [0, r]    */

#define WFN(id) static double w_##id(const struct pl_filter_ctx *f, double x)

WFN(box)       { (void) f; (void) x; return 1.0; }
WFN(triangle)  { return 1.0 - x / f->radius; }
WFN(cosine)    { (void) f; return cos(x); }
WFN(hann)      { (void) f; return 0.5 + 0.5 * cos(M_PI * x); }
WFN(hamming)   { (void) f; return 0.54 + 0.46 * cos(M_PI * x); }
WFN(welch)     { (void) f; return 1.0 - x * x; }

// Modified Bessel function of the first kind, order 0 (power series)
static double series_i0(double x)
{
    const double q = x * x / 4.0;
    double acc = 1.0, term = q;
    for (int k = 2; term > 1e-12; k++) {
        acc += term;
        term *= q / (k * k);
    }
    return acc;
}

WFN(kaiser)
{
    const double alpha = fmax(f->params[0], 0.0);
    const double norm = series_i0(alpha);
    return series_i0(alpha * sqrt(1.0 - x * x)) / norm;
}

WFN(blackman)
{
    const double a = f->params[0];
    const double a0 = (1 - a) / 2.0, a1 = 1 / 2.0, a2 = a / 2.0;
    x *= M_PI;
    return a0 + a1 * cos(x) + a2 * cos(2 * x);
}

WFN(bohman)
{
    (void) f;
    const double px = M_PI * x;
    return (1.0 - x) * cos(px) + sin(px) / M_PI;
}

WFN(gaussian)  { return exp(-2.0 * x * x / f->params[0]); }

WFN(quadratic)
{
    (void) f;
    if (x < 0.5)
        return 1.0 - 4.0/3.0 * (x * x);
    return 2.0 / 3.0 * (x - 1.5) * (x - 1.5);
}

WFN(sinc)
{
    (void) f;
    if (x < 1e-8)
        return 1.0;
    x *= M_PI;
    return sin(x) / x;
}

WFN(jinc)
{
    (void) f;
    if (x < 1e-8)
        return 1.0;
    x *= M_PI;
    return 2.0 * j1(x) / x;
}

WFN(sphinx)
{
    (void) f;
    if (x < 1e-8)
        return 1.0;
    x *= M_PI;
    return 3.0 * (sin(x) - x * cos(x)) / (x * x * x);
}

// Two-parameter (B, C) piecewise cubic (Mitchell–Netravali family)
WFN(cubic)
{
    const double b = f->params[0], c = f->params[1];
    const double p0 = 6.0 - 2.0 * b,
                 p2 = -18.0 + 12.0 * b + 6.0 * c,
                 p3 = 12.0 - 9.0 * b - 6.0 * c,
                 q0 = 8.0 * b + 24.0 * c,
                 q1 = -12.0 * b - 48.0 * c,
                 q2 = 6.0 * b + 30.0 * c,
                 q3 = -b - 6.0 * c;
    if (x < 1.0)
        return (p0 + x * x * (p2 + x * p3)) / p0;
    return (q0 + x * (q1 + x * (q2 + x * q3))) / p0;
}

WFN(spline16)
{
    (void) f;
    if (x < 1.0)
        return ((x - 9.0/5.0 ) * x - 1.0/5.0 ) * x + 1.0;
    return ((-1.0/3.0 * (x-1) + 4.0/5.0) * (x-1) - 7.0/15.0 ) * (x-1);
}

WFN(spline36)
{
    (void) f;
    if (x < 1.0)
        return ((13.0/11.0 * x - 453.0/209.0) * x - 3.0/209.0) * x + 1.0;
    if (x < 2.0)
        return ((-6.0/11.0 * (x-1) + 270.0/209.0) * (x-1) - 156.0/ 209.0) * (x-1);
    return ((1.0/11.0 * (x-2) - 45.0/209.0) * (x-2) +  26.0/209.0) * (x-2);
}

WFN(spline64)
{
    (void) f;
    if (x < 1.0)
        return ((49.0/41.0 * x - 6387.0/2911.0) * x - 3.0/2911.0) * x + 1.0;
    if (x < 2.0)
        return ((-24.0/41.0 * (x-1) + 4032.0/2911.0) * (x-1) - 2328.0/2911.0) * (x-1);
    if (x < 3.0)
        return ((6.0/41.0 * (x-2) - 1008.0/2911.0) * (x-2) + 582.0/2911.0) * (x-2);
    return ((-1.0/41.0 * (x-3) + 168.0/2911.0) * (x-3) - 97.0/2911.0) * (x-3);
}

WFN(zero)      { (void) f; (void) x; return 0.0; }

/* ------------------------------------------------------------------------ */
/* function table                                                            */

#define FN(sym, nm, fn, r, ...) \
    const struct pl_filter_function sym = { .name = nm, .weight = w_##fn, .radius = r, __VA_ARGS__ }

FN(pl_filter_function_box,       "box",       box,       1.0, .resizable = true);
FN(pl_filter_function_triangle,  "triangle",  triangle,  1.0, .resizable = true);
FN(pl_filter_function_cosine,    "cosine",    cosine,    M_PI / 2.0);
FN(pl_filter_function_hann,      "hann",      hann,      1.0);
FN(pl_filter_function_hamming,   "hamming",   hamming,   1.0);
FN(pl_filter_function_welch,     "welch",     welch,     1.0);
FN(pl_filter_function_kaiser,    "kaiser",    kaiser,    1.0, .params = {2.0}, .tunable = {true});
FN(pl_filter_function_blackman,  "blackman",  blackman,  1.0, .params = {0.16}, .tunable = {true});
FN(pl_filter_function_bohman,    "bohman",    bohman,    1.0);
FN(pl_filter_function_gaussian,  "gaussian",  gaussian,  2.0, .resizable = true,
                                                              .params = {1.0}, .tunable = {true});
FN(pl_filter_function_quadratic, "quadratic", quadratic, 1.5);
FN(pl_filter_function_sinc,      "sinc",      sinc,      1.0, .resizable = true);
FN(pl_filter_function_jinc,      "jinc",      jinc,      1.2196698912665045, .resizable = true);
FN(pl_filter_function_sphinx,    "sphinx",    sphinx,    1.4302966531242027, .resizable = true);
FN(pl_filter_function_cubic,     "cubic",     cubic,     2.0, .params = {1.0, 0.0},
                                                              .tunable = {true, true});
FN(pl_filter_function_hermite,   "hermite",   cubic,     1.0, .params = {0.0, 0.0});
FN(pl_filter_function_spline16,  "spline16",  spline16,  2.0);
FN(pl_filter_function_spline36,  "spline36",  spline36,  3.0);
FN(pl_filter_function_spline64,  "spline64",  spline64,  4.0);
FN(pl_filter_function_oversample,"oversample",zero,      0.0, .params = {0.0}, .tunable = {true},
                                                              .opaque = true);

// name aliases kept by the reference's lookup tables
static FN(fn_alias_dirichlet, "dirichlet", box,       1.0, .resizable = true);
static FN(fn_alias_hanning,   "hanning",   hann,      1.0);
static FN(fn_alias_quadric,   "quadric",   quadratic, 1.5);
// (exported objects in the reference: filters.c:505-551)
FN(pl_filter_function_bicubic,  "bicubic",  cubic,     2.0, .params = {1.0, 0.0}, .tunable = {true, true});
FN(pl_filter_function_bcspline, "bcspline", cubic,     2.0, .params = {1.0, 0.0}, .tunable = {true, true});
FN(pl_filter_function_catmull_rom, "catmull_rom", cubic, 2.0, .params = {0.0, 0.5}, .tunable = {true, true});
FN(pl_filter_function_mitchell, "mitchell", cubic,     2.0, .params = {1/3.0, 1/3.0}, .tunable = {true, true});
FN(pl_filter_function_robidoux, "robidoux", cubic,     2.0,
   .params = {12 / (19 + 9 * M_SQRT2), 113 / (58 + 216 * M_SQRT2)}, .tunable = {true, true});
FN(pl_filter_function_robidouxsharp, "robidouxsharp", cubic, 2.0,
   .params = {6 / (13 + 7 * M_SQRT2), 7 / (2 + 12 * M_SQRT2)}, .tunable = {true, true});
#define fn_alias_bicubic       pl_filter_function_bicubic
#define fn_alias_bcspline      pl_filter_function_bcspline
#define fn_alias_catmull       pl_filter_function_catmull_rom
#define fn_alias_mitchell      pl_filter_function_mitchell
#define fn_alias_robidoux      pl_filter_function_robidoux
#define fn_alias_robidouxsharp pl_filter_function_robidouxsharp

const struct pl_filter_function * const pl_filter_functions[] = {
    &pl_filter_function_box,      &fn_alias_dirichlet,
    &pl_filter_function_triangle, &pl_filter_function_cosine,
    &pl_filter_function_hann,     &fn_alias_hanning,
    &pl_filter_function_hamming,  &pl_filter_function_welch,
    &pl_filter_function_kaiser,   &pl_filter_function_blackman,
    &pl_filter_function_bohman,   &pl_filter_function_gaussian,
    &pl_filter_function_quadratic,&fn_alias_quadric,
    &pl_filter_function_sinc,     &pl_filter_function_jinc,
    &pl_filter_function_sphinx,   &pl_filter_function_cubic,
    &pl_filter_function_hermite,  &fn_alias_bicubic,
    &fn_alias_bcspline,           &fn_alias_catmull,
    &fn_alias_mitchell,           &fn_alias_robidoux,
    &fn_alias_robidouxsharp,      &pl_filter_function_spline16,
    &pl_filter_function_spline36, &pl_filter_function_spline64,
    &pl_filter_function_oversample,
    NULL,
};

const int pl_num_filter_functions =
    sizeof(pl_filter_functions) / sizeof(pl_filter_functions[0]) - 1;

const struct pl_filter_function *pl_find_filter_function(const char *name)
{
    for (int i = 0; name && i < pl_num_filter_functions; i++) {
        if (!strcmp(name, pl_filter_functions[i]->name))
            return pl_filter_functions[i];
    }
    return NULL;
}

// the older name -> function table (filters.c:998-1044): every entry of pl_filter_functions but the
// opaque oversampler, under its own name, behind a leading "none"
const struct pl_filter_function_preset pl_filter_function_presets[] = {
    {"none", NULL},
#define P(fn) {(fn).name, &(fn)}
    P(pl_filter_function_box), P(fn_alias_dirichlet), P(pl_filter_function_triangle),
    P(pl_filter_function_cosine), P(pl_filter_function_hann), P(fn_alias_hanning),
    P(pl_filter_function_hamming), P(pl_filter_function_welch), P(pl_filter_function_kaiser),
    P(pl_filter_function_blackman), P(pl_filter_function_bohman), P(pl_filter_function_gaussian),
    P(pl_filter_function_quadratic), P(fn_alias_quadric), P(pl_filter_function_sinc),
    P(pl_filter_function_jinc), P(pl_filter_function_sphinx), P(pl_filter_function_cubic),
    P(pl_filter_function_hermite), P(pl_filter_function_bicubic), P(pl_filter_function_bcspline),
    P(pl_filter_function_catmull_rom), P(pl_filter_function_mitchell), P(pl_filter_function_robidoux),
    P(pl_filter_function_robidouxsharp), P(pl_filter_function_spline16),
    P(pl_filter_function_spline36), P(pl_filter_function_spline64),
#undef P
    {0},
};

const int pl_num_filter_function_presets =
    sizeof(pl_filter_function_presets) / sizeof(pl_filter_function_presets[0]) - 1;

const struct pl_filter_function_preset *pl_find_filter_function_preset(const char *name)
{
    for (int i = 0; name && i < pl_num_filter_function_presets; i++) {
        if (!strcmp(name, pl_filter_function_presets[i].name))
            return &pl_filter_function_presets[i];
    }
    return NULL;
}

/* ------------------------------------------------------------------------ */
/* config table                                                              */

#define JINC_R3 3.2383154841662362076499
#define JINC_R4 4.2410628637960698819573
#define ROBIDOUX_BC      {12 / (19 + 9 * M_SQRT2), 113 / (58 + 216 * M_SQRT2)}
#define ROBIDOUXSHARP_BC {6 / (13 + 7 * M_SQRT2), 7 / (2 + 12 * M_SQRT2)}

#define CFG(sym, nm, desc, kern, ...) \
    const struct pl_filter_config sym = { .name = nm, .description = desc, \
        .kernel = &pl_filter_function_##kern, __VA_ARGS__ }

#define UP   PL_FILTER_UPSCALING
#define DOWN PL_FILTER_DOWNSCALING
#define MIX  PL_FILTER_FRAME_MIXING
#define SCAL PL_FILTER_SCALING
#define ALL  PL_FILTER_ALL

CFG(pl_filter_spline16, "spline16", "Spline (2 taps)", spline16, .allowed = ALL);
CFG(pl_filter_spline36, "spline36", "Spline (3 taps)", spline36, .allowed = ALL);
CFG(pl_filter_spline64, "spline64", "Spline (4 taps)", spline64, .allowed = ALL);
CFG(pl_filter_nearest,  "nearest",  "Nearest neighbor", box, .radius = 0.5,
    .allowed = UP, .recommended = UP);
CFG(pl_filter_box,      "box",      "Box averaging", box, .radius = 0.5,
    .allowed = SCAL, .recommended = DOWN);
CFG(pl_filter_bilinear, "bilinear", "Bilinear", triangle,
    .allowed = ALL, .recommended = SCAL);
static CFG(cfg_linear,  "linear",   "Linear mixing", triangle,
    .allowed = MIX, .recommended = MIX);
static CFG(cfg_triangle,"triangle", NULL, triangle, .allowed = SCAL);
CFG(pl_filter_gaussian, "gaussian", "Gaussian", gaussian, .params = {1.0},
    .allowed = ALL, .recommended = SCAL);
CFG(pl_filter_sinc,     "sinc",     "Sinc (unwindowed)", sinc, .radius = 2.0,
    .allowed = ALL);
CFG(pl_filter_lanczos,  "lanczos",  "Lanczos", sinc,
    .window = &pl_filter_function_sinc, .radius = 3.0,
    .allowed = ALL, .recommended = SCAL);
CFG(pl_filter_ginseng,  "ginseng",  "Ginseng (Jinc-Sinc)", sinc,
    .window = &pl_filter_function_jinc, .radius = 3.0, .allowed = ALL);
CFG(pl_filter_ewa_jinc, "ewa_jinc", "EWA Jinc (unwindowed)", jinc,
    .radius = JINC_R3, .polar = true, .allowed = SCAL);
CFG(pl_filter_ewa_lanczos, "ewa_lanczos", "Jinc (EWA Lanczos)", jinc,
    .window = &pl_filter_function_jinc, .radius = JINC_R3, .polar = true,
    .allowed = SCAL, .recommended = UP);
CFG(pl_filter_ewa_lanczossharp, "ewa_lanczossharp", "Sharpened Jinc", jinc,
    .window = &pl_filter_function_jinc, .radius = JINC_R3,
    .blur = 0.98125058372237073562493, .polar = true,
    .allowed = SCAL, .recommended = UP);
CFG(pl_filter_ewa_lanczos4sharpest, "ewa_lanczos4sharpest",
    "Sharpened Jinc-AR, 4 taps", jinc,
    .window = &pl_filter_function_jinc, .radius = JINC_R4,
    .blur = 0.88451209326050047745788, .antiring = 0.8, .polar = true,
    .allowed = SCAL, .recommended = UP);
CFG(pl_filter_ewa_ginseng, "ewa_ginseng", "EWA Ginseng", jinc,
    .window = &pl_filter_function_sinc, .radius = JINC_R3, .polar = true,
    .allowed = SCAL);
CFG(pl_filter_ewa_hann, "ewa_hann", "EWA Hann", jinc,
    .window = &pl_filter_function_hann, .radius = JINC_R3, .polar = true,
    .allowed = SCAL);
static CFG(cfg_ewa_hanning, "ewa_hanning", NULL, jinc,
    .window = &pl_filter_function_hann, .radius = JINC_R3, .polar = true,
    .allowed = SCAL);
CFG(pl_filter_bicubic,  "bicubic",  "Bicubic", cubic, .params = {1.0, 0.0},
    .allowed = SCAL, .recommended = SCAL);
static CFG(cfg_cubic,   "cubic",    "Cubic", cubic, .params = {1.0, 0.0},
    .allowed = MIX);
CFG(pl_filter_hermite,  "hermite",  "Hermite", hermite,
    .allowed = ALL, .recommended = DOWN | MIX);
CFG(pl_filter_catmull_rom, "catmull_rom", "Catmull-Rom", cubic,
    .params = {0.0, 0.5}, .allowed = ALL, .recommended = SCAL);
CFG(pl_filter_mitchell, "mitchell", "Mitchell-Netravali", cubic,
    .params = {1/3.0, 1/3.0}, .allowed = ALL, .recommended = DOWN);
CFG(pl_filter_mitchell_clamp, "mitchell_clamp", "Mitchell (clamped)", cubic,
    .params = {1/3.0, 1/3.0}, .clamp = 1.0, .allowed = ALL);
CFG(pl_filter_robidoux, "robidoux", "Robidoux", cubic,
    .params = ROBIDOUX_BC, .allowed = ALL);
CFG(pl_filter_robidouxsharp, "robidouxsharp", "RobidouxSharp", cubic,
    .params = ROBIDOUXSHARP_BC, .allowed = ALL);
CFG(pl_filter_ewa_robidoux, "ewa_robidoux", "EWA Robidoux", cubic,
    .params = ROBIDOUX_BC, .polar = true, .allowed = SCAL);
CFG(pl_filter_ewa_robidouxsharp, "ewa_robidouxsharp", "EWA RobidouxSharp", cubic,
    .params = ROBIDOUXSHARP_BC, .polar = true, .allowed = SCAL);
CFG(pl_filter_oversample, "oversample", "Oversampling", oversample,
    .params = {0.0}, .allowed = UP | MIX, .recommended = UP | MIX);

// Same priority order as the reference's list (filters.c:943-976)
const struct pl_filter_config * const pl_filter_configs[] = {
    &pl_filter_bilinear, &cfg_triangle, &cfg_linear, &pl_filter_nearest,
    &pl_filter_spline16, &pl_filter_spline36, &pl_filter_spline64,
    &pl_filter_lanczos, &pl_filter_ewa_lanczos, &pl_filter_ewa_lanczossharp,
    &pl_filter_ewa_lanczos4sharpest, &pl_filter_bicubic, &cfg_cubic,
    &pl_filter_hermite, &pl_filter_gaussian, &pl_filter_oversample,
    &pl_filter_mitchell, &pl_filter_mitchell_clamp, &pl_filter_sinc,
    &pl_filter_ginseng, &pl_filter_ewa_jinc, &pl_filter_ewa_ginseng,
    &pl_filter_ewa_hann, &cfg_ewa_hanning, &pl_filter_catmull_rom,
    &pl_filter_robidoux, &pl_filter_robidouxsharp, &pl_filter_ewa_robidoux,
    &pl_filter_ewa_robidouxsharp,
    NULL,
};

const int pl_num_filter_configs =
    sizeof(pl_filter_configs) / sizeof(pl_filter_configs[0]) - 1;

const struct pl_filter_config *
pl_find_filter_config(const char *name, enum pl_filter_usage usage)
{
    for (int i = 0; name && i < pl_num_filter_configs; i++) {
        const struct pl_filter_config *c = pl_filter_configs[i];
        if ((c->allowed & usage) == usage && !strcmp(name, c->name))
            return c;
    }
    return NULL;
}

// the older name -> config table (filters.h:28-58 COMMON_FILTER_PRESETS, filters.c:1046-1065):
// recommended scalers first, then the rest, then two aliases without a description
const struct pl_filter_preset pl_filter_presets[] = {
    {"none", NULL, "Built-in sampling"},
    PLH_COMMON_FILTER_PRESETS
    {"triangle", &pl_filter_bilinear, NULL},
    {"ewa_hanning", &pl_filter_ewa_hann, NULL},
    {0},
};

const int pl_num_filter_presets = sizeof(pl_filter_presets) / sizeof(pl_filter_presets[0]) - 1;

const struct pl_filter_preset *pl_find_filter_preset(const char *name)
{
    for (int i = 0; name && i < pl_num_filter_presets; i++) {
        if (!strcmp(name, pl_filter_presets[i].name))
            return &pl_filter_presets[i];
    }
    return NULL;
}

/* ------------------------------------------------------------------------ */
/* comparison                                                                */

bool pl_filter_function_eq(const struct pl_filter_function *a,
                           const struct pl_filter_function *b)
{
    return (a ? a->weight : NULL) == (b ? b->weight : NULL);
}

bool pl_filter_config_eq(const struct pl_filter_config *a,
                         const struct pl_filter_config *b)
{
    if (!a || !b)
        return a == b;

    if (!pl_filter_function_eq(a->kernel, b->kernel) ||
        !pl_filter_function_eq(a->window, b->window))
        return false;
    if (a->radius != b->radius || a->clamp != b->clamp || a->blur != b->blur ||
        a->taper != b->taper || a->polar != b->polar || a->antiring != b->antiring)
        return false;

    for (int i = 0; i < PL_FILTER_MAX_PARAMS; i++) {
        if (a->kernel->tunable[i] != b->kernel->tunable[i])
            return false;
        if (a->kernel->tunable[i] && a->params[i] != b->params[i])
            return false;
        if (!a->window)
            continue;
        if (a->window->tunable[i] != b->window->tunable[i])
            return false;
        if (a->window->tunable[i] && a->wparams[i] != b->wparams[i])
            return false;
    }
    return true;
}

/* ------------------------------------------------------------------------ */
/* sampling + LUT generation                                                 */

float pl_filter_radius_bound(const struct pl_filter_config *c)
{
    const float r = c->radius && c->kernel->resizable ? c->radius : c->kernel->radius;
    return c->blur > 0.0 ? r * c->blur : r;
}

static inline struct pl_filter_ctx fn_ctx(const struct pl_filter_function *fn,
                                          const float *user, float radius)
{
    struct pl_filter_ctx ctx = { .radius = radius };
    for (int i = 0; i < PL_FILTER_MAX_PARAMS; i++)
        ctx.params[i] = fn->tunable[i] ? user[i] : fn->params[i];
    return ctx;
}

double pl_filter_sample(const struct pl_filter_config *c, double x)
{
    const float radius = pl_filter_radius_bound(c);
    x = fabs(x);
    if (x > radius)
        return 0.0; // outside the support; kernels are undefined there

    // taper (flat top) and blur stretch the kernel's own coordinate
    double kx = x <= c->taper ? 0.0 : (x - c->taper) / (1.0 - c->taper / radius);
    if (c->blur > 0.0)
        kx /= c->blur;

    const struct pl_filter_ctx kctx = fn_ctx(c->kernel, c->params, radius);
    double k = c->kernel->weight(&kctx, kx);

    // the window is always stretched over the full support
    if (c->window) {
        const struct pl_filter_ctx wctx = fn_ctx(c->window, c->wparams, c->window->radius);
        const double wx = x / radius * c->window->radius;
        k *= c->window->weight(&wctx, wx);
    }

    return k < 0 ? (1 - c->clamp) * k : k;
}

// Locate the outermost |w| = cutoff crossing (-> radius) and the first one
// (-> radius_zero) by scanning [0, bound] in float steps of 0.01 and refining
// each bracket with one secant step. All arithmetic in float, like the
// reference, so that e.g. ewa_lanczos yields exactly 3.159482.
static void scan_cutoffs(const struct pl_filter_config *c, float cutoff,
                         float *radius, float *radius_zero)
{
    const float bound = pl_filter_radius_bound(c);
    const float step = 1e-2f;
    float x0 = 0.0, f0 = pl_filter_sample(c, x0);
    bool any = false;

    for (float x = 0.0; x < bound + step; x += step) {
        const float fx = pl_filter_sample(c, x);
        const bool falling = f0 > cutoff && fx <= cutoff;
        const bool rising  = f0 < -cutoff && fx >= -cutoff;
        if (falling || rising) {
            float root = x - fx * (x - x0) / (fx - f0);
            root = fminf(root, bound);
            *radius = root;
            if (!any)
                *radius_zero = root;
            any = true;
        }
        x0 = x;
        f0 = fx;
    }

    if (!any)
        *radius_zero = *radius = bound;
}

static void fill_row(const struct pl_filter_config *cfg, int row_size,
                     double phase, float *out)
{
    // taps sit at integer positions 0..row_size-1, the sample point at
    // (row_size/2 - 1) + phase
    const double center = (row_size / 2 - 1) + phase;
    double sum = 0.0;
    for (int i = 0; i < row_size; i++) {
        const double w = pl_filter_sample(cfg, i - center);
        out[i] = w;
        sum += w;
    }
    for (int i = 0; i < row_size; i++)
        out[i] /= sum; // energy preservation
}

struct filter_priv {
    struct pl_filter_t pub;
    struct pl_filter_function kernel, window; // owned copies (API lifetime)
    float *weights;
};

pl_filter pl_filter_generate(pl_log log, const struct pl_filter_params *params)
{
    if (!params || params->lut_entries <= 0 || !params->config.kernel) {
        pl_msg(log, PL_LOG_FATAL, "Invalid params: missing lut_entries or config.kernel");
        return NULL;
    }
    if (params->config.kernel->opaque) {
        pl_msg(log, PL_LOG_ERR, "Trying to use opaque kernel '%s' in non-opaque context!",
               params->config.kernel->name);
        return NULL;
    }
    if (params->config.window && params->config.window->opaque) {
        pl_msg(log, PL_LOG_ERR, "Trying to use opaque window '%s' in non-opaque context!",
               params->config.window->name);
        return NULL;
    }

    struct filter_priv *p = calloc(1, sizeof(*p));
    if (!p)
        return NULL;
    struct pl_filter_t *f = &p->pub;
    f->params = *params;
    p->kernel = *params->config.kernel;
    f->params.config.kernel = &p->kernel;
    if (params->config.window) {
        p->window = *params->config.window;
        f->params.config.window = &p->window;
    }

    scan_cutoffs(&params->config, params->cutoff, &f->radius, &f->radius_zero);
    f->radius_cutoff = f->radius;   // legacy alias

    const int n = params->lut_entries;
    if (params->config.polar) {
        // radial LUT: entry i samples x = radius * i / (n-1)
        p->weights = malloc(n * sizeof(float));
        for (int i = 0; i < n; i++) {
            const double x = f->radius * i / (n - 1);
            p->weights[i] = pl_filter_sample(&params->config, x);
        }
    } else {
        f->row_size = ceilf(f->radius) * 2;
        if (params->max_row_size && f->row_size > params->max_row_size) {
            pl_msg(log, PL_LOG_INFO, "Required filter size %d exceeds the maximum "
                   "allowed size of %d. This may result in adverse effects "
                   "(aliasing, or moiré artifacts).", f->row_size, params->max_row_size);
            f->row_size = params->max_row_size;
            f->insufficient = true;
        }
        const int align = params->row_stride_align > 0 ? params->row_stride_align : 1;
        f->row_stride = (f->row_size + align - 1) / align * align;

        // one normalised row per sub-pixel phase i/(n-1)
        p->weights = calloc((size_t) n * f->row_stride, sizeof(float));
        for (int i = 0; i < n; i++) {
            fill_row(&f->params.config, f->row_size, i / (double) (n - 1),
                     p->weights + (size_t) f->row_stride * i);
        }
    }

    f->weights = p->weights;
    return f;
}

void pl_filter_free(pl_filter *filter)
{
    if (!filter || !*filter)
        return;
    struct filter_priv *p = (struct filter_priv *) *filter;
    free(p->weights);
    free(p);
    *filter = NULL;
}
