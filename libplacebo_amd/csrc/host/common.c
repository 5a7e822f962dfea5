/*
 * libplacebo-hip — rect / 3x3 matrix helpers (role of the reference's
 * src/common.c:57-240). Precision conventions are the reference's: float
 * products for apply/mul, a double-precision adjugate for the inverse — they
 * matter because the resulting matrices become kernel constants.
 */
#include <math.h>

#include <libplacebo/common.h>
#include "host_common.h"

void pl_rect2df_normalize(pl_rect2df *rc)
{
    const pl_rect2df in = *rc;
    rc->x0 = PL_MIN(in.x0, in.x1);
    rc->x1 = PL_MAX(in.x0, in.x1);
    rc->y0 = PL_MIN(in.y0, in.y1);
    rc->y1 = PL_MAX(in.y0, in.y1);
}

pl_rect2d pl_rect2df_round(const pl_rect2df *rc)
{
    return (pl_rect2d) {
        .x0 = roundf(rc->x0),
        .x1 = roundf(rc->x1),
        .y0 = roundf(rc->y0),
        .y1 = roundf(rc->y1),
    };
}

const pl_matrix3x3 pl_matrix3x3_identity = {{
    { 1, 0, 0 },
    { 0, 1, 0 },
    { 0, 0, 1 },
}};

const pl_transform3x3 pl_transform3x3_identity = {
    .mat = {{
        { 1, 0, 0 },
        { 0, 1, 0 },
        { 0, 0, 1 },
    }},
};

void pl_matrix3x3_apply(const pl_matrix3x3 *mat, float vec[3])
{
    const float x = vec[0], y = vec[1], z = vec[2];
    for (int i = 0; i < 3; i++)
        vec[i] = mat->m[i][0] * x + mat->m[i][1] * y + mat->m[i][2] * z;
}

void pl_matrix3x3_scale(pl_matrix3x3 *mat, float scale)
{
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            mat->m[i][j] *= scale;
    }
}

void pl_matrix3x3_invert(pl_matrix3x3 *mat)
{
    // inverse = adjugate / determinant, evaluated in double
    double m[3][3], adj[3][3];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            m[i][j] = mat->m[i][j];
    }

    adj[0][0] =  (m[1][1] * m[2][2] - m[2][1] * m[1][2]);
    adj[0][1] = -(m[0][1] * m[2][2] - m[2][1] * m[0][2]);
    adj[0][2] =  (m[0][1] * m[1][2] - m[1][1] * m[0][2]);
    adj[1][0] = -(m[1][0] * m[2][2] - m[2][0] * m[1][2]);
    adj[1][1] =  (m[0][0] * m[2][2] - m[2][0] * m[0][2]);
    adj[1][2] = -(m[0][0] * m[1][2] - m[1][0] * m[0][2]);
    adj[2][0] =  (m[1][0] * m[2][1] - m[2][0] * m[1][1]);
    adj[2][1] = -(m[0][0] * m[2][1] - m[2][0] * m[0][1]);
    adj[2][2] =  (m[0][0] * m[1][1] - m[1][0] * m[0][1]);

    double det = m[0][0] * adj[0][0] + m[1][0] * adj[0][1] + m[2][0] * adj[0][2];
    det = 1.0 / det;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            mat->m[i][j] = det * adj[i][j];
    }
}

void pl_matrix3x3_mul(pl_matrix3x3 *a, const pl_matrix3x3 *b)
{
    const pl_matrix3x3 o = *a;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++)
            a->m[r][c] = o.m[r][0] * b->m[0][c] + o.m[r][1] * b->m[1][c] + o.m[r][2] * b->m[2][c];
    }
}

void pl_matrix3x3_rmul(const pl_matrix3x3 *a, pl_matrix3x3 *b)
{
    pl_matrix3x3 m = *a;
    pl_matrix3x3_mul(&m, b);
    *b = m;
}

void pl_transform3x3_apply(const pl_transform3x3 *t, float vec[3])
{
    pl_matrix3x3_apply(&t->mat, vec);
    for (int i = 0; i < 3; i++)
        vec[i] += t->c[i];
}

void pl_transform3x3_scale(pl_transform3x3 *t, float scale)
{
    pl_matrix3x3_scale(&t->mat, scale);
    for (int i = 0; i < 3; i++)
        t->c[i] *= scale;
}

void pl_transform3x3_invert(pl_transform3x3 *t)
{
    // y = M x + c  =>  x = M^-1 y - M^-1 c
    pl_matrix3x3_invert(&t->mat);
    const float c[3] = { t->c[0], t->c[1], t->c[2] };
    for (int i = 0; i < 3; i++)
        t->c[i] = -(t->mat.m[i][0] * c[0] + t->mat.m[i][1] * c[1] + t->mat.m[i][2] * c[2]);
}

// reference src/common.c:469-500
void pl_rect2df_rotate(pl_rect2df *rc, pl_rotation rot)
{
    rot = pl_rotation_normalize(rot);
    if (!rot)
        return;
    float x0 = rc->x0, y0 = rc->y0, x1 = rc->x1, y1 = rc->y1;
    if (rot >= PL_ROTATION_180) {
        // a half turn swaps both pairs of edges
        rot -= PL_ROTATION_180;
        float t;
        t = x0; x0 = x1; x1 = t;
        t = y0; y0 = y1; y1 = t;
    }
    if (rot == PL_ROTATION_90)
        *rc = (pl_rect2df) { .x0 = y1, .y0 = x0, .x1 = y0, .y1 = x1 };
    else
        *rc = (pl_rect2df) { .x0 = x0, .y0 = y0, .x1 = x1, .y1 = y1 };
}
