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

typedef struct pl_rect3d {
    int x0, y0, z0;
    int x1, y1, z1;
} pl_rect3d;

typedef struct pl_rect2df {
    float x0, y0;
    float x1, y1;
} pl_rect2df;

typedef struct pl_rect3df {
    float x0, y0, z0;
    float x1, y1, z1;
} pl_rect3df;

#define pl_rect_w(r) ((r).x1 - (r).x0)
#define pl_rect_h(r) ((r).y1 - (r).y0)
#define pl_rect_d(r) ((r).z1 - (r).z0)

#define pl_rect2d_eq(a, b) \
    ((a).x0 == (b).x0 && (a).x1 == (b).x1 && (a).y0 == (b).y0 && (a).y1 == (b).y1)
#define pl_rect3d_eq(a, b) \
    (pl_rect2d_eq(a, b) && (a).z0 == (b).z0 && (a).z1 == (b).z1)

// Un-flip: afterwards x0 <= x1 etc.
PL_API void pl_rect2d_normalize(pl_rect2d *rc);
PL_API void pl_rect3d_normalize(pl_rect3d *rc);
PL_API void pl_rect2df_normalize(pl_rect2df *rc);
PL_API void pl_rect3df_normalize(pl_rect3df *rc);
PL_API pl_rect3d pl_rect3df_round(const pl_rect3df *rc);

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
PL_API void pl_matrix3x3_apply_rc(const pl_matrix3x3 *mat, pl_rect3df *rc);
PL_API void pl_transform3x3_apply_rc(const pl_transform3x3 *t, pl_rect3df *rc);

// 2x2 counterparts (row-major), used for temporal dither rotations and rect maths
typedef struct pl_matrix2x2 {
    float m[2][2];
} pl_matrix2x2;

PL_API extern const pl_matrix2x2 pl_matrix2x2_identity;
PL_API pl_matrix2x2 pl_matrix2x2_rotation(float angle);   // radians, counter-clockwise
PL_API void pl_matrix2x2_apply(const pl_matrix2x2 *mat, float vec[2]);
PL_API void pl_matrix2x2_apply_rc(const pl_matrix2x2 *mat, pl_rect2df *rc);
PL_API void pl_matrix2x2_mul(pl_matrix2x2 *a, const pl_matrix2x2 *b);
PL_API void pl_matrix2x2_rmul(const pl_matrix2x2 *a, pl_matrix2x2 *b);
PL_API void pl_matrix2x2_scale(pl_matrix2x2 *mat, float scale);
PL_API void pl_matrix2x2_invert(pl_matrix2x2 *mat);

typedef struct pl_transform2x2 {
    pl_matrix2x2 mat;
    float c[2];
} pl_transform2x2;

PL_API extern const pl_transform2x2 pl_transform2x2_identity;
PL_API void pl_transform2x2_apply(const pl_transform2x2 *t, float vec[2]);
PL_API void pl_transform2x2_apply_rc(const pl_transform2x2 *t, pl_rect2df *rc);
PL_API void pl_transform2x2_mul(pl_transform2x2 *a, const pl_transform2x2 *b);
PL_API void pl_transform2x2_rmul(const pl_transform2x2 *a, pl_transform2x2 *b);
PL_API void pl_transform2x2_scale(pl_transform2x2 *t, float scale);
PL_API void pl_transform2x2_invert(pl_transform2x2 *t);
// Axis-aligned bounding box of a transformed rect
PL_API pl_rect2df pl_transform2x2_bounds(const pl_transform2x2 *t, const pl_rect2df *rc);

// Aspect-ratio helpers for placing an image rect inside a target rect. `panscan` in [0, 1]
// blends between letter-boxing (0) and cropping (1).
PL_API float pl_rect2df_aspect(const pl_rect2df *rc);
PL_API void pl_rect2df_aspect_set(pl_rect2df *rc, float aspect, float panscan);
#define pl_rect2df_aspect_copy(rc, src, panscan) \
    pl_rect2df_aspect_set((rc), pl_rect2df_aspect(src), (panscan))
// Like aspect_copy, but never scales `rc` beyond the size of `src`
PL_API void pl_rect2df_aspect_fit(pl_rect2df *rc, const pl_rect2df *src, float panscan);
// Scale about the centre / translate
PL_API void pl_rect2df_stretch(pl_rect2df *rc, float stretch_x, float stretch_y);
PL_API void pl_rect2df_offset(pl_rect2df *rc, float offset_x, float offset_y);
#define pl_rect2df_zoom(rc, zoom) pl_rect2df_stretch((rc), (zoom), (zoom))

static inline float pl_aspect_rotate(float aspect, pl_rotation rot)
{
    return (rot % PL_ROTATION_180) ? 1.0 / aspect : aspect;
}

#define pl_rect2df_aspect_set_rot(rc, aspect, rot, panscan) \
    pl_rect2df_aspect_set((rc), pl_aspect_rotate((aspect), (rot)), (panscan))
#define pl_rect2df_aspect_copy_rot(rc, src, panscan, rot) \
    pl_rect2df_aspect_set_rot((rc), pl_rect2df_aspect(src), (rot), (panscan))

PL_API_END

#endif // LIBPLACEBO_COMMON_H_
