/*
 * libplacebo-hip: common geometric / matrix types.
 * API-compatible subset of the reference's src/include/libplacebo/common.h
 * (pl_rect2d/pl_rect2df :42-83, pl_matrix3x3 :110-136, pl_transform3x3 :140-165).
 */
#ifndef LIBPLACEBO_COMMON_H_
#define LIBPLACEBO_COMMON_H_

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#include <libplacebo/config.h>

PL_API_BEGIN

// Integer / float 2D rectangles: (x0,y0) top-left, (x1,y1) bottom-right.
// x0 > x1 or y0 > y1 denotes a flipped (mirrored) rectangle.
typedef struct pl_rect2d {
    int x0, y0;
    int x1, y1;
} pl_rect2d;

typedef struct pl_rect2df {
    float x0, y0;
    float x1, y1;
} pl_rect2df;

#define pl_rect_w(r) ((r).x1 - (r).x0)
#define pl_rect_h(r) ((r).y1 - (r).y0)

PL_API void pl_rect2df_normalize(pl_rect2df *rc);

// Rotation in multiples of 90 degrees clockwise (reference common.h:205-231)
typedef int pl_rotation;
enum {
    PL_ROTATION_0 = 0, PL_ROTATION_90 = 1, PL_ROTATION_180 = 2, PL_ROTATION_270 = 3,
    PL_ROTATION_360 = 4,    // equivalent to PL_ROTATION_0; values outside [0, 4) are legal
};

static inline pl_rotation pl_rotation_normalize(pl_rotation rot)
{
    return (rot % PL_ROTATION_360 + PL_ROTATION_360) % PL_ROTATION_360;
}

// Rotates the coordinate system of a rect (PL_ROTATION_90: the x axis becomes the y axis)
PL_API void pl_rect2df_rotate(pl_rect2df *rc, pl_rotation rot);
PL_API pl_rect2d pl_rect2df_round(const pl_rect2df *rc);

// Row-major 3x3 matrix: out[i] = sum_j m[i][j] * in[j]
typedef struct pl_matrix3x3 {
    float m[3][3];
} pl_matrix3x3;

PL_API extern const pl_matrix3x3 pl_matrix3x3_identity;

PL_API void pl_matrix3x3_apply(const pl_matrix3x3 *mat, float vec[3]);
PL_API void pl_matrix3x3_scale(pl_matrix3x3 *mat, float scale);
PL_API void pl_matrix3x3_invert(pl_matrix3x3 *mat);
// mat := mat * b
PL_API void pl_matrix3x3_mul(pl_matrix3x3 *a, const pl_matrix3x3 *b);
// mat := b * mat
PL_API void pl_matrix3x3_rmul(const pl_matrix3x3 *a, pl_matrix3x3 *b);

// Affine transform: out = mat * in + c
typedef struct pl_transform3x3 {
    pl_matrix3x3 mat;
    float c[3];
} pl_transform3x3;

PL_API extern const pl_transform3x3 pl_transform3x3_identity;

PL_API void pl_transform3x3_apply(const pl_transform3x3 *t, float vec[3]);
PL_API void pl_transform3x3_scale(pl_transform3x3 *t, float scale);
PL_API void pl_transform3x3_invert(pl_transform3x3 *t);

PL_API_END

#endif // LIBPLACEBO_COMMON_H_
