/*
 * libplacebo-hip — thin C interface to the HIP runtime used by the C host
 * layer (gpu.c). Everything the `pl_gpu` front-end needs from the device:
 * memory, 2-D copies, stream ordering, event timers.
 */
#ifndef PLH_BACKEND_H_
#define PLH_BACKEND_H_

#include <stddef.h>
#include <stdint.h>

#include "plh_device.h"

#ifdef __cplusplus
extern "C" {
#endif

struct plh_dev_info {
    char name[256];
    char arch[64];
    int compute_units;
    int wavefront_size;
    size_t lds_per_block;
    size_t total_mem;
    int clock_khz;
    uint8_t uuid[16];
    int pci_domain, pci_bus, pci_device;
};

int plh_dev_count(void);
const char *plh_strerror(int err);
int plh_dev_open(int device, struct plh_dev_info *info);
int plh_stream_create(int device, plh_stream *out);
int plh_stream_create_masked(int device, int ncus, plh_stream *out);
// the device that owns `s` (not the calling thread's current one) and, optionally, its CU count
int plh_stream_device(plh_stream s, int *cus);
// raise a kernel's dynamic-LDS limit on the stream's device, once per (kernel, device); `done` =
// the caller's static per-kernel device mask
int plh_kernel_needs_lds(const void *kernel, plh_stream s, size_t bytes, uint64_t *done);
// the PQ transfer pair as piecewise cubics (pqseg.hiph) on the device that owns `s`: built and
// uploaded on first use; consts = { m1, c3, m2, 1 / m2, 1 / m1 }. NULL = not available
const void *plh_pqseg_tables(plh_stream s, const float consts[5]);
void plh_stream_destroy(plh_stream s);
int plh_stream_sync(plh_stream s);
int plh_stream_idle(plh_stream s);   // 1 idle, 0 busy, < 0 error

void *plh_malloc(int device, size_t size);
void plh_free(void *ptr);
void *plh_host_alloc(size_t size); // pinned
void *plh_host_alloc_coherent(size_t size); // pinned, fine-grained, device-mapped
void plh_host_free(void *ptr);

int plh_copy2d_h2d(plh_stream s, void *dst, size_t dpitch, const void *src, size_t spitch,
                   size_t row_bytes, size_t rows);
int plh_copy2d_d2h(plh_stream s, void *dst, size_t dpitch, const void *src, size_t spitch,
                   size_t row_bytes, size_t rows);
int plh_copy2d_d2d(plh_stream s, void *dst, size_t dpitch, const void *src, size_t spitch,
                   size_t row_bytes, size_t rows);
int plh_memset(plh_stream s, void *dst, int value, size_t size);

typedef void *plh_event;
int plh_event_create(plh_event *out);
void plh_event_destroy(plh_event e);
int plh_event_record(plh_event e, plh_stream s);
// An event for the END of the pass launched next on this thread: carried by the launch of the
// pass's last kernel as its stop event where the launcher supports it (devmath.hiph:
// PLH_LAUNCH_LAST). plh_launch_stop_taken() after the launch: 1 = it was, 0 = record it yourself.
void plh_launch_offer_stop(plh_event e);
int plh_launch_stop_taken(void);
int plh_stream_wait_event(plh_stream s, plh_event e);
// 1 = ready, 0 = not yet, <0 error
int plh_event_query(plh_event e);
int plh_event_sync(plh_event e);
// elapsed nanoseconds between two completed events
int plh_event_elapsed_ns(plh_event a, plh_event b, uint64_t *ns);

// fills the whole texture with a constant colour (pl_tex_clear_ex)
int plh_launch_clear(plh_stream s, const struct plh_view *dst, const float color[4]);
int plh_launch_swap_words(plh_stream s, const void *src, void *dst, size_t words, int wordsize);
// fills the whole texture with two-colour tiles (pl_frame_clear_tiles): texel (x, y) takes c0 where
// fract((x + 1/2) * kx) < 1/2 and fract((y + 1/2) * ky) < 1/2 agree, c1 where they differ
int plh_launch_clear_tiles(plh_stream s, const struct plh_view *dst, const float c0[4],
                           const float c1[4], float kx, float ky);
// k_noise.hip: plane[y * stride + x] = pcg3d(x + x0, y + y0, seed).x for x < w, y < h
int plh_launch_white_noise(plh_stream s, float *plane, int stride, int w, int h, int x0, int y0,
                           uint32_t seed);

#ifdef __cplusplus
}
#endif

#endif // PLH_BACKEND_H_
