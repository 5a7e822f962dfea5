/*
 * libplacebo-hip: host-memory -> plane upload helpers.
 * API-compatible with the reference's src/include/libplacebo/utils/upload.h
 * (pl_plane_data :33-96, pl_plane_data_from_mask :101, pl_plane_data_from_comps :108,
 * pl_plane_data_align :120, pl_plane_find_fmt :126, pl_upload_plane :140,
 * pl_recreate_plane :150). Behaviour restated from src/utils/upload.c.
 */
#ifndef LIBPLACEBO_UPLOAD_H_
#define LIBPLACEBO_UPLOAD_H_

#include <stdint.h>

#include <libplacebo/gpu.h>
#include <libplacebo/renderer.h>

PL_API_BEGIN

// Host representation of one image plane. Components are listed in memory order
// (lowest bit offset first); `component_map` gives each one's meaning (0..3 = RGBA / YUVA)
struct pl_plane_data {
    enum pl_fmt_type type;  // must not be UINT / SINT
    int width, height;
    int component_size[4];  // bits per component, 0 = absent
    int component_pad[4];   // ignored bits in front of each component
    int component_map[4];
    size_t pixel_stride;    // bytes between pixels (required)
    size_t row_stride;      // bytes between rows (0 = tightly packed)
    bool swapped;           // samples are in non-native byte order

    const void *pixels;     // host data ...
    pl_buf buf;             // ... or a buffer (exactly one of the two)
    size_t buf_offset;

    void (*callback)(void *priv);
    void *priv;
};

// Fill component_size / _pad / _map from one bit mask per semantic component
PL_API void pl_plane_data_from_mask(struct pl_plane_data *data, uint64_t mask[4]);

// Same from (size, shift) in bits per semantic component; not limited to 64-bit pixels
PL_API void pl_plane_data_from_comps(struct pl_plane_data *data, int size[4], int shift[4]);

// Grow every component to byte boundaries by absorbing its padding, if that can be
// done consistently: returns the resulting bit encoding in `out_bits` (zeroed and
// `false` when it cannot, leaving `data` untouched)
PL_API bool pl_plane_data_align(struct pl_plane_data *data, struct pl_bit_encoding *out_bits);

// The texture format pl_upload_plane would pick, NULL if none; `out_map` (optional)
// receives the semantic index of each texture component (-1 = unused)
PL_API pl_fmt pl_plane_find_fmt(pl_gpu gpu, int out_map[4], const struct pl_plane_data *data);

// (Re)create `*tex` to fit `data`, upload it and describe it as a pl_plane.
// `out_plane->shift_x/y` and `->flipped` are left for the caller.
PL_API bool pl_upload_plane(pl_gpu gpu, struct pl_plane *out_plane,
                            pl_tex *tex, const struct pl_plane_data *data);

// Like pl_upload_plane without the upload: a renderable texture for a target plane
PL_API bool pl_recreate_plane(pl_gpu gpu, struct pl_plane *out_plane,
                              pl_tex *tex, const struct pl_plane_data *data);

PL_API_END

#endif // LIBPLACEBO_UPLOAD_H_
