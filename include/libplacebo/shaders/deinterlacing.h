/*
 * libplacebo-hip: field / deinterlacing descriptions.
 * Types of the reference's src/include/libplacebo/shaders/deinterlacing.h (:30-52, :95-122)
 * that pl_frame and pl_render_params refer to. This backend has no deinterlacer:
 * pl_render_params.deinterlace_params is refused with PL_RENDER_ERR_DEINTERLACING and
 * interlaced frames are shown woven.
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

enum pl_deinterlace_algorithm {
    PL_DEINTERLACE_WEAVE = 0,
    PL_DEINTERLACE_BOB,
    PL_DEINTERLACE_YADIF,
    PL_DEINTERLACE_BWDIF,
    PL_DEINTERLACE_ALGORITHM_COUNT,
};

struct pl_deinterlace_params {
    enum pl_deinterlace_algorithm algo;
    bool skip_spatial_check;
};

PL_API_END

#endif // LIBPLACEBO_SHADERS_DEINTERLACING_H_
