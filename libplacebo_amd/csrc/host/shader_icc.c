/*
 * libplacebo-hip -- the ICC entry points of a build without LittleCMS.
 *
 * ICC colour management in the reference is lcms2 (src/shaders/icc.c:26-800: profile parsing, the
 * detection of primaries / gamma, the 3D LUT sampled through cmsDoTransform). This image has no
 * lcms2, and neither has the reference when it is compiled here (PL_HAVE_LCMS undefined): its five
 * entry points then exist and say so (src/shaders/icc.c:802-836). These are the same five with the
 * same behaviour, so that a caller written against libplacebo links and sees what it would see
 * from such a build: pl_icc_open fails with the reference's message, pl_icc_update clears the
 * object and fails (the message once), pl_icc_close is a no-op, and decode / encode -- which
 * nobody can reach without an object -- fail the shader instead of aborting. The renderer renders
 * a frame that carries a profile from its pl_color_space (renderer.c: note_ignored_members).
 */
#include <stdbool.h>

#include <libplacebo/shaders/icc.h>

#include "host_common.h"
#include "shaders_priv.h"

const struct pl_icc_params pl_icc_default_params = { PL_ICC_DEFAULTS };

static const char no_lcms[] = "libplacebo compiled without LittleCMS 2 support!";

pl_icc_object pl_icc_open(pl_log log, const struct pl_icc_profile *profile,
                          const struct pl_icc_params *params)
{
    pl_msg(log, PL_LOG_ERR, "%s", no_lcms);
    return NULL;
}

void pl_icc_close(pl_icc_object *icc)
{
    if (icc)
        *icc = NULL;    // (there never was one)
}

bool pl_icc_update(pl_log log, pl_icc_object *obj, const struct pl_icc_profile *profile,
                   const struct pl_icc_params *params)
{
    static bool warned;
    if (!warned) {
        pl_msg(log, PL_LOG_ERR, "%s", no_lcms);
        warned = true;
    }
    *obj = NULL;
    return false;
}

void pl_icc_decode(pl_shader sh, pl_icc_object icc, pl_shader_obj *lut_obj,
                   struct pl_color_space *out_csp)
{
    SH_FAIL(sh, "pl_icc_decode: %s", no_lcms);
}

void pl_icc_encode(pl_shader sh, pl_icc_object icc, pl_shader_obj *lut_obj)
{
    SH_FAIL(sh, "pl_icc_encode: %s", no_lcms);
}
