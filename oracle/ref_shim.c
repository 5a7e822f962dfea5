/* TEST INFRASTRUCTURE ONLY. Our glue for oracle/_ref/libplref.so: the reference's
 * utils/upload.c pulls in pl_gpu entry points that the CPU half has no backend for;
 * only its pure functions (pl_plane_data_from_mask/_from_comps/_align) are exercised
 * (tests/test_upload.py), so the GPU entry points resolve to stubs that fail. */
#include <libplacebo/filters.h>
#include <libplacebo/gpu.h>

int plref_shim_version(void) { return 2; }

bool pl_tex_recreate(pl_gpu gpu, pl_tex *tex, const struct pl_tex_params *params) { return false; }
bool pl_tex_upload(pl_gpu gpu, const struct pl_tex_transfer_params *params) { return false; }
size_t pl_tex_transfer_size(const struct pl_tex_transfer_params *par) { return 0; }
pl_buf pl_buf_create(pl_gpu gpu, const struct pl_buf_params *params) { return NULL; }
void pl_buf_destroy(pl_gpu gpu, pl_buf *buf) { }
void pl_buf_copy(pl_gpu gpu, pl_buf dst, size_t dst_offset, pl_buf src, size_t src_offset, size_t size) { }
bool pl_buf_copy_swap(pl_gpu gpu, const struct pl_buf_copy_swap_params *params) { return false; }
