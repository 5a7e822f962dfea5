/*
 * libplacebo-hip — dithering stages: host half of src/shaders/dithering.c.
 *   pl_shader_dither            dithering.c:109-274
 *   pl_shader_error_diffusion   dithering.c:326-527 (see k_errdiff.hip)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <libplacebo/shaders/dithering.h>

#include "shaders_priv.h"
#include "cache_priv.h"

const struct pl_dither_params pl_dither_default_params = { PL_DITHER_DEFAULTS };

struct sh_dither_obj {
    pl_buf lut;
    int size;
    enum pl_dither_method method;
};

static void sh_dither_uninit(pl_gpu gpu, void *ptr)
{
    struct sh_dither_obj *obj = ptr;
    pl_buf_destroy(gpu, &obj->lut);
    memset(obj, 0, sizeof(*obj));
}

static void fill_blue_noise(void *data, void *priv)
{
    pl_generate_blue_noise(data, *(const int *) priv);
}

// Representative gamma of a transfer curve (approx_gamma, dithering.c:77-107)
static float approx_gamma(enum pl_color_transfer trc)
{
    switch (trc) {
    case PL_COLOR_TRC_UNKNOWN:
    case PL_COLOR_TRC_LINEAR:
    case PL_COLOR_TRC_SCRGB:    return 1.0f;
    case PL_COLOR_TRC_PRO_PHOTO:
    case PL_COLOR_TRC_GAMMA18:  return 1.8f;
    case PL_COLOR_TRC_GAMMA20:  return 2.0f;
    case PL_COLOR_TRC_GAMMA24:  return 2.4f;
    case PL_COLOR_TRC_GAMMA26:
    case PL_COLOR_TRC_ST428:    return 2.6f;
    case PL_COLOR_TRC_GAMMA28:  return 2.8f;
    case PL_COLOR_TRC_SRGB:
    case PL_COLOR_TRC_BT_1886:
    case PL_COLOR_TRC_GAMMA22:  return 2.2f;
    case PL_COLOR_TRC_PQ:
    case PL_COLOR_TRC_HLG:
    case PL_COLOR_TRC_V_LOG:
    case PL_COLOR_TRC_S_LOG1:
    case PL_COLOR_TRC_S_LOG2:   return 2.0f;
    case PL_COLOR_TRC_COUNT:    break;
    }
    return 1.0f;
}

void pl_shader_dither(pl_shader sh, int new_depth, pl_shader_obj *dither_state,
                      const struct pl_dither_params *params)
{
    if (!sh_require(sh, PL_SHADER_SIG_COLOR, 0, 0))
        return;

    if (new_depth <= 0 || new_depth > 256) {
        pl_msg(sh->log, PL_LOG_WARN, "Invalid dither depth: %d.. ignoring", new_depth);
        return;
    }

    params = PL_DEF(params, &pl_dither_default_params);
    if (params->lut_size < 0 || params->lut_size > 8) {
        SH_FAIL(sh, "Invalid `lut_size` specified: %d", params->lut_size);
        return;
    }

    sh_describef(sh, "dithering (%d bits)", new_depth);
    enum pl_dither_method method = params->method;
    const bool is_lut = method == PL_DITHER_BLUE_NOISE || method == PL_DITHER_ORDERED_LUT;
    struct sh_dither_obj *obj = NULL;
    int lut_size = 0;

    if (is_lut && !dither_state) {
        pl_msg(sh->log, PL_LOG_WARN, "LUT-based dither method specified but no dither "
               "state object given, falling back to non-LUT based methods.");
        method = PL_DITHER_ORDERED_FIXED;
    } else if (is_lut) {
        obj = SH_OBJ(sh, dither_state, PL_SHADER_OBJ_DITHER, struct sh_dither_obj,
                     sh_dither_uninit);
        lut_size = 1 << PL_DEF(params->lut_size, pl_dither_default_params.lut_size);
        if (obj && (!obj->lut || obj->size != lut_size || obj->method != method)) {
            float *mat = malloc(sizeof(float) * lut_size * lut_size);
            if (!mat)
                return;
            if (method == PL_DITHER_ORDERED_LUT) {
                pl_generate_bayer_matrix(mat, lut_size);
            } else {
                // void-and-cluster is O(size^4): memoised (dithering.c:149,158-159)
                int size_arg = lut_size;
                const bool hit = plh_cache_memoize(plh_gpu_cache(SH_GPU(sh)),
                        (PLH_CACHE_KEY_DITHER ^ (uint64_t) method) * (uint64_t) lut_size, mat,
                        sizeof(float) * lut_size * lut_size, fill_blue_noise, &size_arg);
                pl_msg(sh->log, PL_LOG_DEBUG, hit ? "Re-using cached dither matrix (%dx%d)"
                       : "Generated dither matrix (%dx%d)", lut_size, lut_size);
            }
            // The matrix is followed by its transpose: a kernel whose lanes own a COLUMN of output
            // pixels (k_polar_mx: eight rows per lane) reads its dither values as two 16-byte loads
            // from there instead of eight gathers (matrix[size * size + x * size + y] = matrix[y * size + x]).
            const size_t n = (size_t) lut_size * lut_size;
            float *both = malloc(2 * n * sizeof(float));
            if (!both) {
                free(mat);
                return;
            }
            memcpy(both, mat, n * sizeof(float));
            for (int y = 0; y < lut_size; y++) {
                for (int x = 0; x < lut_size; x++)
                    both[n + (size_t) x * lut_size + y] = mat[(size_t) y * lut_size + x];
            }
            pl_buf_destroy(SH_GPU(sh), &obj->lut);
            obj->lut = pl_buf_create(SH_GPU(sh), pl_buf_params(
                .size = 2 * n * sizeof(float), .storable = true, .initial_data = both));
            free(both);
            free(mat);
            obj->size = lut_size;
            obj->method = method;
        }
        if (!obj || !obj->lut) {
            obj = NULL;
            method = PL_DITHER_ORDERED_FIXED; // the reference's fallback (:168-170)
        }
    }

    const bool white = method == PL_DITHER_WHITE_NOISE;
    const int size = obj ? lut_size : 16;
    struct plh_op *op = sh_op(sh, PLH_OP_DITHER);
    if (!op)
        return;
    // i1: 0 = LUT, 1 = ordered-fixed bit tricks, 2 = white noise (i0 = PRNG seed: the frame
    // index if temporal, else 0 -- sh_prng, shaders.c:985-990)
    op->i0 = white ? (params->temporal ? (int32_t) sh->params.index : 0) : size;
    op->i1 = white ? 2 : obj ? 0 : 1;
    op->i2 = !white && params->temporal;
    op->f[0] = (float) ((1LLU << new_depth) - 1);
    op->f[1] = approx_gamma(params->transfer);
    op->f[2] = 1.0f / (float) size;     // GLSL constant expression `1.0/size`
    op->f[3] = new_depth;
    op->f[8] = 1.0f / op->f[0];         // GLSL constant expression `1.0 / scale`
    if (op->i2) {
        const int phase = sh->params.index % 8;
        const float r = phase * (M_PI / 2);
        const float m = phase < 4 ? 1 : -1;
        // the reference uploads mat[2][2] = {{cos r, -sin r}, {sin r * m, cos r * m}}
        // as a column-major mat2: column 0 = (cos r, -sin r), column 1 = (sin r*m, cos r*m)
        // => (rot * pos).x = c0.x*pos.x + c1.x*pos.y, .y = c0.y*pos.x + c1.y*pos.y
        op->f[4] = cos(r);  op->f[5] = sin(r) * m;
        op->f[6] = -sin(r); op->f[7] = cos(r) * m;
    }
    if (obj) {
        op->ptr = pl_hip_buf_ptr(obj->lut);
        op->ptr2 = (const float *) op->ptr + (size_t) size * size;     // (its transpose: above)
        sh_hold(sh, *dither_state);
    }
    sh_listf(sh, "dither(depth=%d, method=%d, size=%d, gamma=%g, temporal=%d, index=%d)\n",
             new_depth, (int) method, white ? 0 : size, op->f[1], (int) params->temporal,
             (int) sh->params.index);
}

// Right-most column (after the (y, x) -> (y, x + y*shift) skew) that the
// current column spills error into (dithering.c:294-311)
static int rightmost_shifted_column(const struct pl_error_diffusion_kernel *k)
{
    int ret = 0;
    for (int y = 0; y <= PL_EDF_MAX_DY; y++) {
        for (int x = PL_EDF_MIN_DX; x <= PL_EDF_MAX_DX; x++) {
            if (k->pattern[y][x - PL_EDF_MIN_DX])
                ret = PL_MAX(ret, x + y * k->shift);
        }
    }
    return ret;
}

size_t pl_error_diffusion_shmem_req(const struct pl_error_diffusion_kernel *kernel, int height)
{
    // ring buffer: (height + MAX_DY) rows x (rightmost + 1) columns of one
    // packed-RGB uint each (dithering.c:313-324)
    const int rows = height + PL_EDF_MAX_DY;
    const int cols = rightmost_shifted_column(kernel) + 1;
    return (size_t) rows * cols * sizeof(uint32_t);
}

bool pl_shader_error_diffusion(pl_shader sh, const struct pl_error_diffusion_params *params)
{
    if (!params->input_tex || !params->output_tex) {
        SH_FAIL(sh, "pl_shader_error_diffusion: missing input/output texture");
        return false;
    }
    const int width = params->input_tex->params.w, height = params->input_tex->params.h;
    const struct pl_glsl_version glsl = sh_glsl(sh);
    const struct pl_error_diffusion_kernel *kernel =
        PL_DEF(params->kernel, &pl_error_diffusion_sierra_lite);

    if (params->output_tex->params.w != width || params->output_tex->params.h != height) {
        SH_FAIL(sh, "pl_shader_error_diffusion: input and output sizes differ");
        return false;
    }
    if (!params->output_tex->params.storable) {
        SH_FAIL(sh, "pl_shader_error_diffusion: output texture must be storable");
        return false;
    }
    if (!sh_require(sh, PL_SHADER_SIG_NONE, width, height))
        return false;
    if (params->new_depth <= 0 || params->new_depth > 256) {
        pl_msg(sh->log, PL_LOG_WARN, "Invalid dither depth: %d.. ignoring", params->new_depth);
        return false;
    }

    // one workgroup walks the sheared columns (dithering.c:352-378)
    const int shifted_width = width + (height - 1) * kernel->shift;
    const int block_size = PL_MIN((int) glsl.max_group_threads, height);
    const int blocks = (height * shifted_width + block_size - 1) / block_size;
    const int ring_rows = height + PL_EDF_MAX_DY;
    const int ring_cols = rightmost_shifted_column(kernel) + 1;
    const size_t shmem_req = (size_t) ring_rows * ring_cols * sizeof(uint32_t);
    if (!sh_try_compute(sh, block_size, 1, false, shmem_req)) {
        pl_msg(sh->log, PL_LOG_ERR, "Cannot execute error diffusion kernel: insufficient "
               "compute shader memory (%zu bytes)!", shmem_req);
        sh->failed = true;
        return false;
    }

    struct plh_errdiff_args *a = calloc(1, sizeof(*a));
    if (!a) {
        sh->failed = true;
        return false;
    }
    plh_tex_view(params->input_tex, &a->src);
    plh_tex_view(params->output_tex, &a->dst);
    a->width = width;
    a->height = height;
    a->quant = (1 << params->new_depth) - 1;
    a->shift = kernel->shift;
    a->divisor = kernel->divisor;
    memcpy(a->pattern, kernel->pattern, sizeof(a->pattern));
    a->ring_rows = ring_rows;
    a->ring_cols = ring_cols;
    a->block_size = block_size;
    a->blocks = blocks;

    free(sh->errdiff);
    sh->errdiff = a;
    sh->kind = PLH_SHADER_ERROR_DIFFUSION;
    sh->output = PL_SHADER_SIG_NONE;
    sh_describef(sh, "error diffusion (%s, %d bits)", kernel->name, params->new_depth);
    sh_listf(sh, "error_diffusion(kernel=%s, depth=%d, %dx%d, block=%d, steps=%d, "
             "ring=%dx%d (%zu B LDS))\n", kernel->name, params->new_depth, width, height,
             block_size, blocks, ring_rows, ring_cols, shmem_req);
    return true;
}
