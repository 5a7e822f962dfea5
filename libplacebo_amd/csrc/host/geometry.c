/*
 * libplacebo-hip — rect placement and 2x2 / 3-D rect helpers of common.h that the render
 * path itself does not use but applications placing an image inside a target do
 * (semantics: the reference's src/common.c:57-98, 115-130, 245-470).
 */
#include <math.h>

#include <libplacebo/common.h>
#include "host_common.h"

static inline void order_i(int *lo, int *hi)
{
    if (*lo > *hi) {
        const int t = *lo;
        *lo = *hi;
        *hi = t;
    }
}

static inline void order_f(float *lo, float *hi)
{
    // written with min/max so that a NaN edge behaves like the reference's PL_MIN/PL_MAX
    const float a = *lo, b = *hi;
    *lo = PL_MIN(a, b);
    *hi = PL_MAX(a, b);
}

void pl_rect2d_normalize(pl_rect2d *rc)
{
    order_i(&rc->x0, &rc->x1);
    order_i(&rc->y0, &rc->y1);
}

void pl_rect3d_normalize(pl_rect3d *rc)
{
    order_i(&rc->x0, &rc->x1);
    order_i(&rc->y0, &rc->y1);
    order_i(&rc->z0, &rc->z1);
}

void pl_rect3df_normalize(pl_rect3df *rc)
{
    order_f(&rc->x0, &rc->x1);
    order_f(&rc->y0, &rc->y1);
    order_f(&rc->z0, &rc->z1);
}

pl_rect3d pl_rect3df_round(const pl_rect3df *rc)
{
    pl_rect3d out;
    out.x0 = roundf(rc->x0);
    out.y0 = roundf(rc->y0);
    out.z0 = roundf(rc->z0);
    out.x1 = roundf(rc->x1);
    out.y1 = roundf(rc->y1);
    out.z1 = roundf(rc->z1);
    return out;
}

/* ---- a rect is two corner points: transform both ---- */

void pl_matrix3x3_apply_rc(const pl_matrix3x3 *mat, pl_rect3df *rc)
{
    float lo[3] = { rc->x0, rc->y0, rc->z0 };
    float hi[3] = { rc->x1, rc->y1, rc->z1 };
    pl_matrix3x3_apply(mat, lo);
    pl_matrix3x3_apply(mat, hi);
    *rc = (pl_rect3df) { lo[0], lo[1], lo[2], hi[0], hi[1], hi[2] };
}

void pl_transform3x3_apply_rc(const pl_transform3x3 *t, pl_rect3df *rc)
{
    float lo[3] = { rc->x0, rc->y0, rc->z0 };
    float hi[3] = { rc->x1, rc->y1, rc->z1 };
    pl_transform3x3_apply(t, lo);
    pl_transform3x3_apply(t, hi);
    *rc = (pl_rect3df) { lo[0], lo[1], lo[2], hi[0], hi[1], hi[2] };
}

/* ---- 2x2 ---- */

const pl_matrix2x2 pl_matrix2x2_identity = {{ { 1, 0 }, { 0, 1 } }};
const pl_transform2x2 pl_transform2x2_identity = { .mat = {{ { 1, 0 }, { 0, 1 } }} };

pl_matrix2x2 pl_matrix2x2_rotation(float angle)
{
    const float c = cosf(angle), s = sinf(angle);
    return (pl_matrix2x2) {{ { c, -s }, { s, c } }};
}

void pl_matrix2x2_apply(const pl_matrix2x2 *mat, float vec[2])
{
    const float x = vec[0], y = vec[1];
    vec[0] = mat->m[0][0] * x + mat->m[0][1] * y;
    vec[1] = mat->m[1][0] * x + mat->m[1][1] * y;
}

void pl_matrix2x2_apply_rc(const pl_matrix2x2 *mat, pl_rect2df *rc)
{
    float lo[2] = { rc->x0, rc->y0 }, hi[2] = { rc->x1, rc->y1 };
    pl_matrix2x2_apply(mat, lo);
    pl_matrix2x2_apply(mat, hi);
    *rc = (pl_rect2df) { lo[0], lo[1], hi[0], hi[1] };
}

void pl_matrix2x2_mul(pl_matrix2x2 *a, const pl_matrix2x2 *b)
{
    const pl_matrix2x2 o = *a;
    for (int r = 0; r < 2; r++) {
        for (int c = 0; c < 2; c++)
            a->m[r][c] = o.m[r][0] * b->m[0][c] + o.m[r][1] * b->m[1][c];
    }
}

void pl_matrix2x2_rmul(const pl_matrix2x2 *a, pl_matrix2x2 *b)
{
    pl_matrix2x2 m = *a;
    pl_matrix2x2_mul(&m, b);
    *b = m;
}

void pl_matrix2x2_scale(pl_matrix2x2 *mat, float scale)
{
    for (int r = 0; r < 2; r++) {
        mat->m[r][0] *= scale;
        mat->m[r][1] *= scale;
    }
}

void pl_matrix2x2_invert(pl_matrix2x2 *mat)
{
    const float a = mat->m[0][0], b = mat->m[0][1], c = mat->m[1][0], d = mat->m[1][1];
    const float rcp = 1.0f / (a * d - b * c);
    *mat = (pl_matrix2x2) {{ { d * rcp, -b * rcp }, { -c * rcp, a * rcp } }};
}

void pl_transform2x2_apply(const pl_transform2x2 *t, float vec[2])
{
    pl_matrix2x2_apply(&t->mat, vec);
    vec[0] += t->c[0];
    vec[1] += t->c[1];
}

void pl_transform2x2_apply_rc(const pl_transform2x2 *t, pl_rect2df *rc)
{
    float lo[2] = { rc->x0, rc->y0 }, hi[2] = { rc->x1, rc->y1 };
    pl_transform2x2_apply(t, lo);
    pl_transform2x2_apply(t, hi);
    *rc = (pl_rect2df) { lo[0], lo[1], hi[0], hi[1] };
}

// a := a o b (apply b first): x -> A (B x + cb) + ca
void pl_transform2x2_mul(pl_transform2x2 *a, const pl_transform2x2 *b)
{
    float c[2] = { b->c[0], b->c[1] };
    pl_matrix2x2_apply(&a->mat, c);
    a->c[0] += c[0];
    a->c[1] += c[1];
    pl_matrix2x2_mul(&a->mat, &b->mat);
}

void pl_transform2x2_rmul(const pl_transform2x2 *a, pl_transform2x2 *b)
{
    pl_transform2x2 m = *a;
    pl_transform2x2_mul(&m, b);
    *b = m;
}

void pl_transform2x2_scale(pl_transform2x2 *t, float scale)
{
    pl_matrix2x2_scale(&t->mat, scale);
    t->c[0] *= scale;
    t->c[1] *= scale;
}

void pl_transform2x2_invert(pl_transform2x2 *t)
{
    // y = M x + c  =>  x = M^-1 y - M^-1 c
    pl_matrix2x2_invert(&t->mat);
    float c[2] = { t->c[0], t->c[1] };
    pl_matrix2x2_apply(&t->mat, c);
    t->c[0] = -c[0];
    t->c[1] = -c[1];
}

pl_rect2df pl_transform2x2_bounds(const pl_transform2x2 *t, const pl_rect2df *rc)
{
    pl_rect2df box = { INFINITY, INFINITY, -INFINITY, -INFINITY };
    for (int corner = 0; corner < 4; corner++) {
        float p[2] = { (corner & 1) ? rc->x1 : rc->x0, (corner & 2) ? rc->y1 : rc->y0 };
        pl_transform2x2_apply(t, p);
        box.x0 = fminf(box.x0, p[0]);
        box.x1 = fmaxf(box.x1, p[0]);
        box.y0 = fminf(box.y0, p[1]);
        box.y1 = fmaxf(box.y1, p[1]);
    }
    return box;
}

/* ---- aspect-ratio placement ---- */

float pl_rect2df_aspect(const pl_rect2df *rc)
{
    const float w = fabsf(pl_rect_w(*rc)), h = fabsf(pl_rect_h(*rc));
    return h ? w / h : 0.0f;
}

void pl_rect2df_stretch(pl_rect2df *rc, float stretch_x, float stretch_y)
{
    const float cx = (rc->x0 + rc->x1) / 2.0, cy = (rc->y0 + rc->y1) / 2.0;
    rc->x0 = rc->x0 * stretch_x + cx * (1.0 - stretch_x);
    rc->x1 = rc->x1 * stretch_x + cx * (1.0 - stretch_x);
    rc->y0 = rc->y0 * stretch_y + cy * (1.0 - stretch_y);
    rc->y1 = rc->y1 * stretch_y + cy * (1.0 - stretch_y);
}

void pl_rect2df_aspect_set(pl_rect2df *rc, float aspect, float panscan)
{
    const float have = pl_rect2df_aspect(rc);
    if (!(aspect > 0) || !have || aspect == have)
        return;
    // ratio > 1: the rect has to become wider relative to its height. With panscan = 0 the
    // short axis shrinks (letter-box), with panscan = 1 the long axis grows (crop).
    if (aspect > have) {
        const float ratio = aspect / have;
        pl_rect2df_stretch(rc, powf(ratio, panscan), powf(ratio, panscan - 1.0));
    } else {
        const float ratio = have / aspect;
        pl_rect2df_stretch(rc, powf(ratio, panscan - 1.0), powf(ratio, panscan));
    }
}

void pl_rect2df_aspect_fit(pl_rect2df *rc, const pl_rect2df *src, float panscan)
{
    const float w = fabs(pl_rect_w(*rc)), h = fabs(pl_rect_h(*rc));
    if (!w || !h)
        return;
    const float sx = fabs(pl_rect_w(*src)) / w, sy = fabs(pl_rect_h(*src)) / h;
    if (sx > 1.0 || sy > 1.0) {
        // `src` does not fit at its own size: fall back to matching the aspect ratio
        pl_rect2df_aspect_copy(rc, src, panscan);
    } else {
        pl_rect2df_stretch(rc, sx, sy);
    }
}

void pl_rect2df_offset(pl_rect2df *rc, float offset_x, float offset_y)
{
    // offsets are in image direction: flipped rects move the other way
    const float dx = rc->x1 < rc->x0 ? -offset_x : offset_x;
    const float dy = rc->y1 < rc->y0 ? -offset_y : offset_y;
    rc->x0 += dx;
    rc->x1 += dx;
    rc->y0 += dy;
    rc->y1 += dy;
}
