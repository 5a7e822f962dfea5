/*
 * libplacebo-hip: host plane description -> texture format -> upload.
 *
 * Restates the behaviour of the reference's src/utils/upload.c:
 *   pl_plane_data_from_comps  :46-76     pl_plane_data_from_mask :78-95
 *   pl_plane_data_align       :97-163    pl_plane_find_fmt       :165-224
 *   pl_upload_plane           :226-330   pl_recreate_plane       :332-376
 * Known answers: src/tests/utils.c:9-98 (tests/test_upload.py).
 *
 * Endian-swapped data is swapped on the host before the copy (the reference uses a
 * compute-shader swap, :268-323): the bytes that reach the texture are the same.
 */
#include <stdlib.h>
#include <string.h>

#include <libplacebo/utils/upload.h>

#include "host_common.h"

struct bitfield { int sem, size, shift; };

// used fields first, then by position in the pixel
static int bitfield_order(const void *pa, const void *pb)
{
    const struct bitfield *a = pa, *b = pb;
    if (!a->size != !b->size)
        return a->size ? -1 : 1;
    return (a->shift > b->shift) - (a->shift < b->shift);
}

void pl_plane_data_from_comps(struct pl_plane_data *data, int size[4], int shift[4])
{
    struct bitfield f[4];
    for (int c = 0; c < 4; c++)
        f[c] = (struct bitfield) { c, size[c], shift[c] };
    qsort(f, 4, sizeof(f[0]), bitfield_order);

    int cursor = 0; // first bit not yet accounted for
    for (int c = 0; c < 4; c++) {
        const bool used = f[c].size > 0;
        data->component_size[c] = used ? f[c].size : 0;
        data->component_pad[c]  = used ? f[c].shift - cursor : 0;
        data->component_map[c]  = used ? f[c].sem : 0;
        if (used)
            cursor = f[c].shift + f[c].size;
    }
}

void pl_plane_data_from_mask(struct pl_plane_data *data, uint64_t mask[4])
{
    int size[4], shift[4];
    for (int c = 0; c < 4; c++) {
        size[c]  = __builtin_popcountll(mask[c]);
        shift[c] = mask[c] ? __builtin_ctzll(mask[c]) : 0;
        // (masks must be contiguous runs of bits)
    }
    pl_plane_data_from_comps(data, size, shift);
}

bool pl_plane_data_align(struct pl_plane_data *data, struct pl_bit_encoding *out_bits)
{
    struct pl_plane_data d = *data;
    int depth = 0, shift = 0, sample = 0;
    int pos = 0;    // bit position where the current component's padding starts

    for (int c = 0; c < 4 && d.component_size[c]; c++) {
        const int orig = d.component_size[c];
        // eat padding in front, down to the previous byte boundary (= a left shift)
        const int start = pos + d.component_pad[c];
        int front = start & 7;
        if (front > d.component_pad[c])
            front = d.component_pad[c];
        // eat padding behind, up to the next byte boundary (= trailing zero bits); the
        // last component may always grow
        const int end = start + orig;
        int back = (8 - (end & 7)) & 7;
        const bool last = c == 3 || !d.component_size[c + 1];
        if (!last) {
            if (back > d.component_pad[c + 1])
                back = d.component_pad[c + 1];
            d.component_pad[c + 1] -= back;
        }
        d.component_pad[c] -= front;
        d.component_size[c] = orig + front + back;

        // every component must agree on the encoding
        if (c == 0) {
            depth = orig; shift = front; sample = d.component_size[c];
        } else if (depth != orig || shift != front || sample != d.component_size[c]) {
            goto fail;
        }
        pos += d.component_pad[c] + d.component_size[c];
    }

    if (d.pixel_stride && (size_t) pos > d.pixel_stride * 8)
        goto fail;

    *data = d;
    if (out_bits) {
        *out_bits = (struct pl_bit_encoding) {
            .sample_depth = sample, .color_depth = depth, .bit_shift = shift,
        };
    }
    return true;

fail:
    if (out_bits)
        *out_bits = (struct pl_bit_encoding) {0};
    return false;
}

pl_fmt pl_plane_find_fmt(pl_gpu gpu, int out_map[4], const struct pl_plane_data *data)
{
    int scratch[4];
    if (!out_map)
        out_map = scratch;

    int ncomp = 0;
    for (int c = 0; c < 4; c++) {
        if (data->component_size[c])
            ncomp = c + 1;
    }

    for (int n = 0; n < gpu->num_formats; n++) {
        pl_fmt fmt = gpu->formats[n];
        if (fmt->opaque || fmt->num_components < ncomp || fmt->type != data->type ||
            fmt->texel_size != data->pixel_stride || !(fmt->caps & PL_FMT_CAP_SAMPLEABLE))
            continue;

        // walk the texture's components: padding occupies a whole (unused) one
        int map[4] = { -1, -1, -1, -1 };
        int tc = 0;
        bool fits = true;
        for (int c = 0; c < ncomp && fits; c++) {
            if (data->component_pad[c]) {
                fits = tc < 4 && fmt->host_bits[tc] == data->component_pad[c];
                tc++;
            }
            if (fits && data->component_size[c]) {
                fits = tc < 4 && fmt->host_bits[tc] == data->component_size[c];
                if (fits)
                    map[tc++] = data->component_map[c];
            }
        }
        if (!fits)
            continue;

        if (data->row_stride % fmt->texel_align) {
            pl_msg(gpu->log, PL_LOG_WARN, "Format '%s' rejected: row stride %zu is not a multiple "
                   "of its texel alignment %zu (API usage bug?)", fmt->name, data->row_stride,
                   fmt->texel_align);
            continue;
        }
        memcpy(out_map, map, sizeof(map));
        return fmt;
    }

    for (int c = 0; c < 4; c++)
        out_map[c] = -1;
    return NULL;
}

static void describe_plane(struct pl_plane *out, pl_tex tex, const int map[4])
{
    if (!out)
        return;
    out->texture = tex;
    out->components = 0;
    for (int c = 0; c < 4; c++) {
        out->component_mapping[c] = map[c];
        if (map[c] >= 0)
            out->components = c + 1;
    }
}

bool pl_upload_plane(pl_gpu gpu, struct pl_plane *out_plane, pl_tex *tex,
                     const struct pl_plane_data *data)
{
    if (!data->buf == !data->pixels) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_upload_plane: exactly one of `pixels` and `buf` must be set");
        return false;
    }

    int map[4];
    pl_fmt fmt = pl_plane_find_fmt(gpu, map, data);
    if (!fmt) {
        pl_msg(gpu->log, PL_LOG_ERR, "Failed picking any compatible texture format for a plane!");
        return false;
    }

    if (!pl_tex_recreate(gpu, tex, pl_tex_params(
            .w = data->width, .h = data->height, .format = fmt,
            .sampleable = true, .host_writable = true,
            // (beyond the reference: every HIP texture can be read back)
            .host_readable = !!(fmt->caps & PL_FMT_CAP_HOST_READABLE),
            .blit_src = !!(fmt->caps & PL_FMT_CAP_BLITTABLE))))
    {
        pl_msg(gpu->log, PL_LOG_ERR, "Failed initializing plane texture!");
        return false;
    }
    describe_plane(out_plane, *tex, map);

    const size_t pitch = PL_DEF(data->row_stride, (size_t) data->width * fmt->texel_size);
    struct pl_tex_transfer_params tp = {
        .tex = *tex, .row_pitch = pitch,
        .ptr = (void *) data->pixels, .buf = data->buf, .buf_offset = data->buf_offset,
        .callback = data->callback, .priv = data->priv,
    };
    if (!data->swapped)
        return pl_tex_upload(gpu, &tp);

    // non-native byte order: swap every sample into a staging copy
    const size_t word = fmt->texel_size / fmt->num_components;
    const int rows = PL_MAX(data->height, 1);       // 1D planes have height 0
    const size_t bytes = pitch * (size_t) (rows - 1) + (size_t) data->width * fmt->texel_size;
    uint8_t *tmp = malloc(bytes);
    if (!tmp)
        return false;
    if (data->pixels) {
        memcpy(tmp, data->pixels, bytes);
    } else if (!pl_buf_read(gpu, data->buf, data->buf_offset, tmp, bytes)) {
        free(tmp);
        return false;
    }
    for (int y = 0; y < rows; y++) {
        uint8_t *row = tmp + pitch * (size_t) y;
        for (size_t i = 0; i + word <= (size_t) data->width * fmt->texel_size; i += word) {
            for (size_t k = 0; k < word / 2; k++) {
                const uint8_t t = row[i + k];
                row[i + k] = row[i + word - 1 - k];
                row[i + word - 1 - k] = t;
            }
        }
    }
    tp.ptr = tmp; tp.buf = NULL; tp.buf_offset = 0;
    const bool ok = pl_tex_upload(gpu, &tp);
    pl_gpu_finish(gpu); // the staging copy must outlive the transfer
    free(tmp);
    return ok;
}

bool pl_recreate_plane(pl_gpu gpu, struct pl_plane *out_plane, pl_tex *tex,
                       const struct pl_plane_data *data)
{
    if (data->swapped) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_recreate_plane does not support non-native endian "
               "plane data (only pl_upload_plane does)");
        return false;
    }

    int map[4];
    pl_fmt fmt = pl_plane_find_fmt(gpu, map, data);
    if (!fmt) {
        pl_msg(gpu->log, PL_LOG_ERR, "Failed picking any compatible texture format for a plane!");
        return false;
    }

    if (!pl_tex_recreate(gpu, tex, pl_tex_params(
            .w = data->width, .h = data->height, .format = fmt,
            .renderable = true,
            .host_readable = !!(fmt->caps & PL_FMT_CAP_HOST_READABLE),
            .blit_dst = !!(fmt->caps & PL_FMT_CAP_BLITTABLE),
            .storable = !!(fmt->caps & PL_FMT_CAP_STORABLE))))
    {
        pl_msg(gpu->log, PL_LOG_ERR, "Failed initializing plane texture!");
        return false;
    }
    describe_plane(out_plane, *tex, map);
    return true;
}
