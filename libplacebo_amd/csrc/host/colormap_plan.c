/*
 * libplacebo-hip -- what a colour-mapping request resolves to, as ONE value.
 *
 * pl_shader_color_map_ex (reference: src/shaders/colorspace.c:1612-1790) decides, before it
 * emits anything, which tone curve and which gamut mapper run between which luminance ranges and
 * whether either can be skipped or folded away. Here that decision is a pure function from the
 * request to a `struct plh_colormap_plan`; shader_color.c only records what the plan says.
 * tests/test_colormap_plan.py holds it to the same decision taken with the reference's own
 * build (oracle/_ref) and to tests/colormap_ref.py::resolve, for every tone / gamut function.
 */
#include <math.h>
#include <string.h>

#include <libplacebo/shaders/colorspace.h>

#include "shaders_priv.h"

struct luma_span { float lo, hi, avg; };

// nominal luminance range of a colour space in `unit`, as seen through `source` metadata
static struct luma_span luma_span(const struct pl_color_space *csp, enum pl_hdr_metadata_type source,
                                  enum pl_hdr_scaling unit)
{
    struct luma_span s = {0};
    pl_color_space_nominal_luma_ex(pl_nominal_luma_params(
        .color = csp, .metadata = source, .scaling = unit,
        .out_min = &s.lo, .out_max = &s.hi, .out_avg = &s.avg,
    ));
    return s;
}

// `v`, or `to` when the two are closer than a millionth (float difference, compared as double)
static float snapped(float v, float to)
{
    return fabs(v - to) < 1e-6 ? to : v;
}

static struct pl_tone_map_params tone_request(const struct pl_color_map_params *par,
                                              const struct pl_color_space *src,
                                              const struct pl_color_space *dst)
{
    const struct luma_span in = luma_span(src, par->metadata, PL_HDR_PQ);
    const struct luma_span out = luma_span(dst, PL_HDR_METADATA_HDR10, PL_HDR_PQ);
    struct pl_tone_map_params t = {
        .function = par->tone_mapping_function ? par->tone_mapping_function : &pl_tone_map_clip,
        .param = par->tone_mapping_param,
        .constants = par->tone_constants,
        .input_scaling = PL_HDR_PQ, .output_scaling = PL_HDR_PQ,
        .lut_size = par->lut_size ? par->lut_size : pl_color_map_default_params.lut_size,
        .input_min = in.lo, .input_max = in.hi, .input_avg = in.avg,
        .output_min = out.lo, .output_max = out.hi,
        .hdr = src->hdr,
    };
    pl_tone_map_params_infer(&t);
    // end points that coincide up to rounding are made to coincide; and a curve never brightens
    // unless inverse tone mapping was asked for
    t.output_max = snapped(t.output_max, t.input_max);
    t.output_min = snapped(t.output_min, t.input_min);
    if (!par->inverse_tone_mapping && t.output_max > t.input_max)
        t.output_max = t.input_max;
    return t;
}

static struct pl_gamut_map_params gamut_request(const struct pl_color_map_params *par,
                                                const struct pl_color_space *src,
                                                const struct pl_color_space *dst)
{
    const struct luma_span out = luma_span(dst, PL_HDR_METADATA_HDR10, PL_HDR_PQ);
    struct pl_gamut_map_params g = {
        .function = par->gamut_mapping ? par->gamut_mapping : &pl_gamut_map_clip,
        .constants = par->gamut_constants,
        .input_gamut = src->hdr.prim, .output_gamut = dst->hdr.prim,
        .min_luma = out.lo, .max_luma = out.hi,
        .lut_stride = 3,
    };
    int *const dims[3] = { &g.lut_size_I, &g.lut_size_C, &g.lut_size_h };
    for (int k = 0; k < 3; k++)
        *dims[k] = par->lut3d_size[k] ? par->lut3d_size[k] : pl_color_map_default_params.lut3d_size[k];
    // a mapper that works in both directions would also EXPAND a smaller source gamut; unless
    // that is wanted, the target is first cut down to what the source can hold
    if (g.function->bidirectional && !par->gamut_expansion &&
        pl_primaries_compatible(&g.input_gamut, &g.output_gamut))
        g.output_gamut = pl_primaries_clip(&g.output_gamut, &g.input_gamut);
    return g;
}

void plh_colormap_resolve(struct plh_colormap_plan *plan, const struct pl_color_map_params *params,
                          const struct pl_color_space *src, const struct pl_color_space *dst,
                          bool stateful)
{
    const struct pl_color_map_params *par = params ? params : &pl_color_map_default_params;
    memset(plan, 0, sizeof(*plan));
    plan->src = *src;
    plan->dst = *dst;
    pl_color_space_infer_map(&plan->src, &plan->dst);
    plan->identity = pl_color_space_equal(&plan->src, &plan->dst);
    if (plan->identity)
        return;

    plan->tone = tone_request(par, &plan->src, &plan->dst);
    plan->gamut = gamut_request(par, &plan->src, &plan->dst);

    // Without a state object there is nowhere to keep a LUT: the curve degrades to its linear
    // stand-in and the gamut mapper to the one that is a matrix (`clip` is closed-form anyway).
    plan->closed_form = !stateful || !par->force_tone_mapping_lut;
    if (!stateful) {
        if (plan->tone.function != &pl_tone_map_clip)
            plan->tone.function = &pl_tone_map_linear;
        if (plan->gamut.function != &pl_gamut_map_clip)
            plan->gamut.function = &pl_gamut_map_saturation;
    }

    plan->need_tone = !pl_tone_map_params_noop(&plan->tone);
    plan->need_gamut = !pl_gamut_map_params_noop(&plan->gamut);
    plan->tone_direct = plan->need_tone && plan->closed_form &&
                        (plan->tone.function == &pl_tone_map_clip ||
                         plan->tone.function == &pl_tone_map_linear);
    // `saturation` is linear in LMS: it becomes part of the output matrix
    plan->fold_saturation = plan->need_gamut && plan->closed_form &&
                            plan->gamut.function == &pl_gamut_map_saturation;
    if (plan->fold_saturation)
        plan->need_gamut = false;
}

// for tests/test_colormap_plan.py
PL_API void plh_test_colormap_resolve(struct plh_colormap_plan *plan, const struct pl_color_map_params *params,
                                      const struct pl_color_space *src, const struct pl_color_space *dst,
                                      bool stateful);
void plh_test_colormap_resolve(struct plh_colormap_plan *plan, const struct pl_color_map_params *params,
                               const struct pl_color_space *src, const struct pl_color_space *dst,
                               bool stateful)
{
    plh_colormap_resolve(plan, params, src, dst, stateful);
}
