/* TEST INFRASTRUCTURE ONLY. Our glue for oracle/_ref/libplref.so: the reference's
 * utils/upload.c pulls in pl_gpu entry points that the CPU half has no backend for;
 * only its pure functions (pl_plane_data_from_mask/_from_comps/_align) are exercised
 * (tests/test_upload.py), so the GPU entry points resolve to stubs that fail. */
#include <libplacebo/filters.h>
#include <libplacebo/gpu.h>

int plref_shim_version(void) { return 3; }

bool pl_tex_recreate(pl_gpu gpu, pl_tex *tex, const struct pl_tex_params *params) { return false; }
bool pl_tex_upload(pl_gpu gpu, const struct pl_tex_transfer_params *params) { return false; }
size_t pl_tex_transfer_size(const struct pl_tex_transfer_params *par) { return 0; }
pl_buf pl_buf_create(pl_gpu gpu, const struct pl_buf_params *params) { return NULL; }
void pl_buf_destroy(pl_gpu gpu, pl_buf *buf) { }
void pl_buf_copy(pl_gpu gpu, pl_buf dst, size_t dst_offset, pl_buf src, size_t src_offset, size_t size) { }
bool pl_buf_copy_swap(pl_gpu gpu, const struct pl_buf_copy_swap_params *params) { return false; }

/* ---- utils/frame_queue.c: host-only logic, compiled as it lies. It touches the GPU only to
 * recycle textures a `map` callback created; the traces in tests/test_frame_queue.py create none. */
#include <stddef.h>
#include <libplacebo/renderer.h>

void pl_tex_destroy(pl_gpu gpu, pl_tex *tex) { if (tex) *tex = NULL; }
void pl_tex_invalidate(pl_gpu gpu, pl_tex tex) { }

// a pl_gpu good enough for pl_queue_create (it only reads `log`)
pl_gpu plref_fake_gpu(void)
{
    static struct pl_gpu_t gpu;
    return &gpu;
}

// where the fields the queue fills in live inside the reference's struct pl_frame
void plref_frame_layout(int out[6])
{
    out[0] = sizeof(struct pl_frame);
    out[1] = offsetof(struct pl_frame, user_data);
    out[2] = offsetof(struct pl_frame, field);
    out[3] = offsetof(struct pl_frame, first_field);
    out[4] = offsetof(struct pl_frame, prev);
    out[5] = offsetof(struct pl_frame, next);
}
