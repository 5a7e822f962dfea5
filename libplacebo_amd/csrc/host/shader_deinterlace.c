/*
 * libplacebo-hip -- pl_shader_deinterlace: records the deinterlacing sampler
 * (reference src/shaders/deinterlacing.c:26-370; the kernel is csrc/hip/k_deinterlace.hip).
 */
#include <libplacebo/shaders/deinterlacing.h>

#include "shaders_priv.h"

const struct pl_deinterlace_params pl_deinterlace_default_params = { PL_DEINTERLACE_DEFAULTS };

static bool same_shape(pl_tex a, pl_tex b)
{
    return a->params.w == b->params.w && a->params.h == b->params.h;
}

void pl_shader_deinterlace(pl_shader sh, const struct pl_deinterlace_source *src,
                           const struct pl_deinterlace_params *params)
{
    params = PL_DEF(params, &pl_deinterlace_default_params);
    pl_tex cur = src->cur.top;
    if (!cur) {
        SH_FAIL(sh, "pl_shader_deinterlace: no current frame");
        return;
    }
    if (sh->pass.s.type != PLH_SAMPLE_NONE || sh->output != PL_SHADER_SIG_NONE) {
        SH_FAIL(sh, "Illegal sequence of shader operations: a sampling stage must "
                "be the first stage of a shader");
        return;
    }
    if ((unsigned) params->algo >= PL_DEINTERLACE_ALGORITHM_COUNT) {
        SH_FAIL(sh, "pl_shader_deinterlace: invalid algorithm %d", (int) params->algo);
        return;
    }
    if (!sh_require(sh, PL_SHADER_SIG_NONE, cur->params.w, cur->params.h))
        return;

    uint8_t mask = PL_DEF(src->component_mask, 0xFu);
    mask &= (1u << cur->params.format->num_components) - 1u;
    if (!mask) {
        SH_FAIL(sh, "pl_shader_deinterlace: empty component mask?");
        return;
    }
    // the neighbours are read with the current frame's geometry (the reference asserts, :97-111)
    if ((src->prev.top && !same_shape(src->prev.top, cur)) ||
        (src->next.top && !same_shape(src->next.top, cur)))
    {
        SH_FAIL(sh, "pl_shader_deinterlace: the previous / next frame must have the size of "
                "the current one");
        return;
    }
    if (!sh_bind(sh, cur, PL_TEX_ADDRESS_MIRROR, NULL))
        return;
    sh_describef(sh, "deinterlacing");

    const enum pl_field first_field = PL_DEF(src->first_field, PL_FIELD_TOP);
    bool intra_only = params->algo != PL_DEINTERLACE_YADIF;
    if (params->algo == PL_DEINTERLACE_BWDIF) {
        intra_only = (!src->prev.top && src->field == first_field) ||
                     (!src->next.top && src->field != first_field);
    }
    pl_tex prev = !intra_only && src->prev.top ? src->prev.top : cur;
    pl_tex next = !intra_only && src->next.top ? src->next.top : cur;

    struct plh_pass *pass = &sh->pass;
    pass->s.type = PLH_SAMPLE_DEINTERLACE;
    pass->s.comp_mask = mask;
    pass->s.scale = 1.0f;
    pass->s.linear = false;
    plh_tex_view(prev, &pass->deint.prev);
    plh_tex_view(next, &pass->deint.next);
    pass->deint.algo = params->algo;
    pass->deint.keep = src->field == PL_FIELD_NONE ? -1 : src->field == PL_FIELD_TOP ? 0 : 1;
    pass->deint.first = src->field == first_field;
    pass->deint.intra_only = intra_only;
    pass->deint.skip_spatial_check = params->skip_spatial_check;
    // "1 unit of brightness on an 8-bit scale", as the shader text carries it ("%f", :129-134)
    pass->deint.spatial_bias = plh_fmtf(1 / 255.0f);
    // (the neighbouring frames are textures the application filled on the main stream, where
    // this pass runs: the two-stream bookkeeping of gpu_hip.c has nothing to order for them)

    static const char *const names[] = { "weave", "bob", "yadif", "bwdif" };
    sh_listf(sh, "deinterlace(%s%s, field=%d, first=%d, mask=0x%x%s)\n", names[params->algo],
             intra_only && params->algo == PL_DEINTERLACE_BWDIF ? " intra" : "",
             (int) src->field, (int) first_field, mask,
             params->skip_spatial_check ? ", no spatial check" : "");
}
