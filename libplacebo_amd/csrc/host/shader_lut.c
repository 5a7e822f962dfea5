/*
 * libplacebo-hip: custom LUTs -- .cube parser and the CUSTOM_LUT op.
 *
 * Restates the behaviour of the reference's src/shaders/lut.c:
 *   pl_lut_parse_cube   :30-185   (header keywords, domain rescaling, body)
 *   fill_lut            :187-210  (RGB -> RGBA texels)
 *   pl_shader_custom_lut :212-280 (shaper matrices, 1D per channel / 3D lookup)
 *   sh_lut linear 1D    :731-745, tetrahedral 3D :762-809 (device: colormap.hiph)
 * Known answers: src/tests/lut.c (tests/test_lut.py).
 */
#include <ctype.h>
#include <stdlib.h>
#include <string.h>

#include <libplacebo/shaders/lut.h>

#include <libplacebo/hip.h>

#include "gpu_priv.h"
#include "shaders_priv.h"

/* ---- .cube parser ---------------------------------------------------------------------- */

struct cursor { const char *p, *end; };

static void skip_space(struct cursor *c)
{
    while (c->p < c->end && isspace((unsigned char) *c->p))
        c->p++;
}

// one line without its terminator, stripped on both sides; advances past the newline
static void next_line(struct cursor *c, const char **start, size_t *len)
{
    const char *s = c->p, *e = s;
    while (e < c->end && *e != '\n')
        e++;
    c->p = e < c->end ? e + 1 : e;
    while (s < e && isspace((unsigned char) *s))
        s++;
    while (e > s && isspace((unsigned char) e[-1]))
        e--;
    *start = s;
    *len = e - s;
}

static bool eat_keyword(const char **s, size_t *len, const char *kw)
{
    const size_t n = strlen(kw);
    if (*len < n || memcmp(*s, kw, n) != 0)
        return false;
    *s += n;
    *len -= n;
    while (*len && isspace((unsigned char) **s)) {
        (*s)++;
        (*len)--;
    }
    return true;
}

// strtod on a bounded, not NUL-terminated token
static bool parse_number(const char *s, size_t len, double *out)
{
    char buf[64];
    if (!len || len >= sizeof(buf))
        return false;
    memcpy(buf, s, len);
    buf[len] = 0;
    char *end;
    *out = strtod(buf, &end);
    return end == buf + len;
}

static bool parse_triple(const char *s, size_t len, float out[3])
{
    for (int i = 0; i < 3; i++) {
        while (len && isspace((unsigned char) *s)) {
            s++; len--;
        }
        size_t n = 0;
        while (n < len && !isspace((unsigned char) s[n]))
            n++;
        double v;
        if (!parse_number(s, n, &v))
            return false;
        out[i] = v;
        s += n;
        len -= n;
    }
    return true;
}

struct pl_custom_lut *pl_lut_parse_cube(pl_log log, const char *str, size_t str_len)
{
    struct pl_custom_lut *lut = calloc(1, sizeof(*lut));
    if (!lut)
        return NULL;
    uint64_t h = 0xcbf29ce484222325ull; // FNV-1a of the file
    for (size_t i = 0; i < str_len; i++)
        h = (h ^ (uint8_t) str[i]) * 0x100000001b3ull;
    lut->signature = h;

    struct cursor c = { str, str + str_len };
    float min[3] = { 0.0f, 0.0f, 0.0f }, max[3] = { 1.0f, 1.0f, 1.0f };
    long entries = 0;
    float *data = NULL;

    // header: everything up to the first line that starts with a number
    for (;;) {
        skip_space(&c);
        if (c.p >= c.end)
            break;
        const char ch = *c.p;
        if (isdigit((unsigned char) ch) || ch == '-')   // isnumeric(), lut.c:24-27
            break;
        const char *s;
        size_t len;
        next_line(&c, &s, &len);
        if (!len)
            continue;
        if (eat_keyword(&s, &len, "TITLE")) {
            pl_msg(log, PL_LOG_INFO, "Loading LUT: %.*s", (int) len, s);
        } else if (len >= 11 && (!memcmp(s, "LUT_3D_SIZE", 11) || !memcmp(s, "LUT_1D_SIZE", 11))) {
            const bool three_d = s[4] == '3';
            eat_keyword(&s, &len, three_d ? "LUT_3D_SIZE" : "LUT_1D_SIZE");
            double v;
            if (!parse_number(s, len, &v) || v != (long) v) {
                pl_msg(log, PL_LOG_ERR, "Failed parsing dimension '%.*s'", (int) len, s);
                goto error;
            }
            const long size = (long) v;
            if (three_d) {
                if (size <= 0 || size > 1024) {
                    pl_msg(log, PL_LOG_ERR, "Invalid 3DLUT size: %ldx%ldx%ld", size, size, size);
                    goto error;
                }
                lut->size[0] = lut->size[1] = lut->size[2] = size;
                entries = size * size * size;
            } else {
                if (size <= 0 || size > 65536) {
                    pl_msg(log, PL_LOG_ERR, "Invalid 1DLUT size: %ld", size);
                    goto error;
                }
                lut->size[0] = size;
                lut->size[1] = lut->size[2] = 0;
                entries = size;
            }
        } else if (eat_keyword(&s, &len, "DOMAIN_MIN")) {
            if (!parse_triple(s, len, min)) {
                pl_msg(log, PL_LOG_ERR, "Failed parsing domain: '%.*s'", (int) len, s);
                goto error;
            }
        } else if (eat_keyword(&s, &len, "DOMAIN_MAX")) {
            if (!parse_triple(s, len, max)) {
                pl_msg(log, PL_LOG_ERR, "Failed parsing domain: '%.*s'", (int) len, s);
                goto error;
            }
        } else if (s[0] == '#') {
            pl_msg(log, PL_LOG_DEBUG, "Unhandled .cube comment: %.*s", (int) len - 1, s + 1);
        } else {
            pl_msg(log, PL_LOG_WARN, "Unhandled .cube line: %.*s", (int) len, s);
        }
    }

    if (!entries) {
        pl_msg(log, PL_LOG_ERR, "Missing LUT size specification?");
        goto error;
    }
    for (int i = 0; i < 3; i++) {
        if (max[i] - min[i] < 1e-6) {
            pl_msg(log, PL_LOG_ERR, "Invalid domain range: [%f, %f]", min[i], max[i]);
            goto error;
        }
    }

    data = malloc(sizeof(float[3]) * entries);
    if (!data)
        goto error;
    for (long n = 0; n < entries; n++) {
        for (int ch = 0; ch < 3; ch++) {
            // a run of "0123456789.-+e", then whitespace
            const char *s = c.p;
            while (c.p < c.end && strchr("0123456789.-+e", *c.p))
                c.p++;
            if (c.p == s) {
                if (c.p >= c.end)
                    pl_msg(log, PL_LOG_ERR, "Failed parsing LUT: Unexpected EOF, expected %ld "
                           "entries, got %ld", entries * 3, n * 3 + ch + 1);
                else
                    pl_msg(log, PL_LOG_ERR, "Failed parsing LUT: Unexpected '%c', expected digit",
                           *c.p);
                goto error;
            }
            double v;
            if (!parse_number(s, c.p - s, &v)) {
                pl_msg(log, PL_LOG_ERR, "Failed parsing float value '%.*s'", (int) (c.p - s), s);
                goto error;
            }
            // rescale to 0.0 - 1.0 (float arithmetic, like the reference)
            const float num = v;
            data[3 * n + ch] = (num - min[ch]) / (max[ch] - min[ch]);
            skip_space(&c);
        }
    }
    skip_space(&c);
    if (c.p < c.end)
        pl_msg(log, PL_LOG_WARN, "Extra data after LUT?... ignoring '%c'", *c.p);
    lut->data = data;
    return lut;

error:
    free(data);
    free(lut);
    return NULL;
}

void pl_lut_free(struct pl_custom_lut **lut)
{
    if (!lut || !*lut)
        return;
    free((void *) (*lut)->data);
    free(*lut);
    *lut = NULL;
}

/* ---- pl_shader_custom_lut ------------------------------------------------------------- */

struct sh_custom_lut_obj {
    uint64_t signature;
    int size[3];
    pl_buf buf;     // rgba32f texels (fill_lut :187-210: alpha = 0)
};

static void custom_lut_uninit(pl_gpu gpu, void *priv)
{
    struct sh_custom_lut_obj *obj = priv;
    pl_buf_destroy(gpu, &obj->buf);
    memset(obj, 0, sizeof(*obj));
}

static void record_matrix(pl_shader sh, const pl_matrix3x3 *m, const char *what)
{
    static const pl_matrix3x3 zero = {0};
    if (!memcmp(m, &zero, sizeof(zero)))
        return;
    struct plh_op *op = sh_op(sh, PLH_OP_AFFINE);
    if (!op)
        return;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            op->f[3 * i + j] = m->m[i][j];
    }
    sh_listf(sh, "%s: color.rgb = M * color.rgb\n", what);
}

void pl_shader_custom_lut(pl_shader sh, const struct pl_custom_lut *lut, pl_shader_obj *lut_state)
{
    if (!lut)
        return;
    int dims;
    if (lut->size[0] > 0 && lut->size[1] > 0 && lut->size[2] > 0) {
        dims = 3;
    } else if (lut->size[0] > 0 && !lut->size[1] && !lut->size[2]) {
        dims = 1;
    } else {
        SH_FAIL(sh, "Invalid dimensions %dx%dx%d for pl_custom_lut, must be 1D or 3D!",
                lut->size[0], lut->size[1], lut->size[2]);
        return;
    }
    if (!sh_require(sh, PL_SHADER_SIG_COLOR, 0, 0))
        return;

    pl_gpu gpu = sh->params.gpu;
    struct sh_custom_lut_obj *obj = SH_OBJ(sh, lut_state, PL_SHADER_OBJ_LUT,
                                           struct sh_custom_lut_obj, custom_lut_uninit);
    if (!obj || !gpu) {
        SH_FAIL(sh, "pl_shader_custom_lut: failed generating LUT object");
        return;
    }
    const size_t n = (size_t) lut->size[0] * PL_DEF(lut->size[1], 1) * PL_DEF(lut->size[2], 1);
    if (!obj->buf || obj->signature != lut->signature || memcmp(obj->size, lut->size, sizeof(obj->size))) {
        float *texels = malloc(n * 4 * sizeof(float));
        if (!texels) {
            SH_FAIL(sh, "pl_shader_custom_lut: out of memory");
            return;
        }
        for (size_t i = 0; i < n; i++) {
            texels[4 * i + 0] = lut->data[3 * i + 0];
            texels[4 * i + 1] = lut->data[3 * i + 1];
            texels[4 * i + 2] = lut->data[3 * i + 2];
            texels[4 * i + 3] = 0.0f;
        }
        pl_buf_destroy(gpu, &obj->buf);
        obj->buf = pl_buf_create(gpu, pl_buf_params(.size = n * 4 * sizeof(float), .storable = true,
                                                    .initial_data = texels));
        free(texels);
        if (!obj->buf) {
            SH_FAIL(sh, "pl_shader_custom_lut: failed generating LUT object");
            return;
        }
        obj->signature = lut->signature;
        memcpy(obj->size, lut->size, sizeof(obj->size));
    }

    record_matrix(sh, &lut->shaper_in, "shaper_in");
    struct plh_op *op = sh_op(sh, PLH_OP_CUSTOM_LUT);
    if (!op)
        return;
    op->i0 = lut->size[0];
    op->i1 = lut->size[1];
    op->i2 = lut->size[2];
    op->ptr = pl_hip_buf_ptr(obj->buf);
    sh_hold(sh, *lut_state);
    sh_describef(sh, dims == 3 ? "custom 3DLUT" : "custom 1DLUT");
    sh_listf(sh, "custom_lut(%dx%dx%d, %s)\n", lut->size[0], lut->size[1], lut->size[2],
             dims == 3 ? "tetrahedral" : "linear");
    record_matrix(sh, &lut->shaper_out, "shaper_out");
}
