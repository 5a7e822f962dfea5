/*
 * libplacebo-hip: deinterlacing (reference src/include/libplacebo/shaders/deinterlacing.h,
 * src/shaders/deinterlacing.c). One HIP kernel, a lane per row pair (csrc/hip/k_deinterlace.hip).
 */
#ifndef LIBPLACEBO_SHADERS_DEINTERLACING_H_
#define LIBPLACEBO_SHADERS_DEINTERLACING_H_

#include <libplacebo/shaders.h>

PL_API_BEGIN

// Field of an interlaced picture
enum pl_field {
    PL_FIELD_NONE = 0, // progressive
    PL_FIELD_EVEN,     // "top" field, even rows
    PL_FIELD_ODD,      // "bottom" field, odd rows
    PL_FIELD_TOP = PL_FIELD_EVEN,
    PL_FIELD_BOTTOM = PL_FIELD_ODD,
};

static inline enum pl_field pl_field_other(enum pl_field field)
{
    return field == PL_FIELD_EVEN ? PL_FIELD_ODD : field == PL_FIELD_ODD ? PL_FIELD_EVEN : field;
}

// Both fields woven into one texture (`top`); separate field textures do not exist upstream either
struct pl_field_pair {
    pl_tex top;
};

#define pl_field_pair(...) ((struct pl_field_pair) { __VA_ARGS__ })

struct pl_deinterlace_source {
    // the frame to deinterlace and its neighbours in time (optional; same size as `cur`)
    struct pl_field_pair prev, cur, next;
    enum pl_field field;        // the field to show: its rows pass through (NONE: the frame as is)
    enum pl_field first_field;  // which field of a frame comes first in time (NONE = top)
    uint8_t component_mask;     // components to process (0 = all the texture has)
};

#define pl_deinterlace_source(...) (&(struct pl_deinterlace_source) { __VA_ARGS__ })

enum pl_deinterlace_algorithm {
    PL_DEINTERLACE_WEAVE = 0,   // both fields as they are
    PL_DEINTERLACE_BOB,         // the shown field's rows doubled
    PL_DEINTERLACE_YADIF,       // edge-directed spatial prediction, limited by the temporal neighbours
    PL_DEINTERLACE_BWDIF,       // motion-adaptive choice between two cubic filters
    PL_DEINTERLACE_ALGORITHM_COUNT,
};

// whether the algorithm reads the previous / next frame
static inline bool pl_deinterlace_needs_refs(enum pl_deinterlace_algorithm algo)
{
    return algo >= PL_DEINTERLACE_YADIF;
}

struct pl_deinterlace_params {
    enum pl_deinterlace_algorithm algo;
    bool skip_spatial_check;    // YADIF only
};

#define PL_DEINTERLACE_DEFAULTS \
    .algo   = PL_DEINTERLACE_BWDIF,

#define pl_deinterlace_params(...) (&(struct pl_deinterlace_params) { PL_DEINTERLACE_DEFAULTS __VA_ARGS__ })
PL_API extern const struct pl_deinterlace_params pl_deinterlace_default_params;

// A sampling stage (the first of a shader): the output has the size of `src->cur`.
PL_API void pl_shader_deinterlace(pl_shader sh, const struct pl_deinterlace_source *src,
                                  const struct pl_deinterlace_params *params);

PL_API_END

#endif // LIBPLACEBO_SHADERS_DEINTERLACING_H_
