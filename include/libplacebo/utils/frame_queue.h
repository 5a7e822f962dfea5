/*
 * libplacebo-hip: (frame, pts) stream -> pl_frame_mix.
 * API-compatible with the reference's src/include/libplacebo/utils/frame_queue.h
 * (pl_queue_status :37, pl_source_frame :44-102, pl_queue_create/destroy/reset :105-113,
 * pl_queue_push :123, pl_queue_push_block :132, pl_queue_params :135-210, pl_queue_update :216,
 * estimates / num_frames / pts_offset / peek :221-241). Behaviour restated from
 * src/utils/frame_queue.c; tests/test_frame_queue.py replays random push / update traces through
 * this implementation and the reference's, compiled as it lies, and compares every output.
 *
 * Thread-safety: safe (pushes may come from a decoder thread while another thread updates).
 */
#ifndef LIBPLACEBO_FRAME_QUEUE_H_
#define LIBPLACEBO_FRAME_QUEUE_H_

#include <stdint.h>

#include <libplacebo/gpu.h>
#include <libplacebo/renderer.h>

PL_API_BEGIN

typedef struct pl_queue_t *pl_queue;

enum pl_queue_status {
    PL_QUEUE_OK,       // success
    PL_QUEUE_EOF,      // the stream has ended
    PL_QUEUE_MORE,     // more frames are needed but not available (yet)
    PL_QUEUE_ERR = -1, // mapping a frame or fetching one failed
};

struct pl_source_frame {
    double pts;      // seconds since the first frame, monotonically increasing
    float duration;  // optional; seeds the frame-rate estimate (recommended when interlaced)

    // != PL_FIELD_NONE: interlaced. The picture becomes two timeline entries (fields), the
    // second half-way to the next picture, each with `prev` / `next` picture references.
    enum pl_field first_field;

    void *frame_data; // opaque, for the callbacks

    // Called when (and only when) the frame is needed on the GPU. `tex` points at four
    // queue-owned texture slots the callback may (re)create its planes in (pl_upload_plane,
    // pl_tex_recreate). A failed map is not retried and `discard` is not called for it.
    bool (*map)(pl_gpu gpu, pl_tex *tex, const struct pl_source_frame *src,
                struct pl_frame *out_frame);
    // Optional: the queue is done with a mapped frame
    void (*unmap)(pl_gpu gpu, struct pl_frame *frame, const struct pl_source_frame *src);
    // Optional: the frame left the queue without ever being mapped
    void (*discard)(const struct pl_source_frame *src);
};

PL_API pl_queue pl_queue_create(pl_gpu gpu);
PL_API void pl_queue_destroy(pl_queue *queue);

// Drop every queued frame and all timing state; allocations and recycled textures are kept
PL_API void pl_queue_reset(pl_queue queue);

// Feed one frame (NULL = end of stream). May be combined with `pl_queue_params.get_frame`.
PL_API void pl_queue_push(pl_queue queue, const struct pl_source_frame *frame);

// As pl_queue_push, but waits up to `timeout` ns while more than a small number of not yet
// mapped frames are already queued. False = timed out, the frame was not taken.
PL_API bool pl_queue_push_block(pl_queue queue, uint64_t timeout,
                                const struct pl_source_frame *frame);

struct pl_queue_params {
    double pts;                    // time of the vsync being rendered; monotonically increasing
    float radius;                  // pl_frame_mix_radius() of the mixer in use
    float vsync_duration;          // hint; the true value is measured from successive `pts`
    float drift_compensation;      // snap `pts` to a frame closer than this, remember the offset
    float interpolation_threshold; // |fps / vps - 1| at or below this: show single frames
    uint64_t timeout;              // ns to wait for a pushed frame (ignored with `get_frame`)

    // Optional pull source. May block; returns OK (frame written), EOF, MORE or ERR.
    enum pl_queue_status (*get_frame)(struct pl_source_frame *out_frame,
                                      const struct pl_queue_params *params);
    void *priv;
};

#define PL_QUEUE_DEFAULTS               \
    .drift_compensation      = 1e-3f,   \
    .interpolation_threshold = 1e-6f,

#define pl_queue_params(...) (&(struct pl_queue_params) { PL_QUEUE_DEFAULTS __VA_ARGS__ })

// Move the queue to `params->pts`: frames too far in the past are unmapped / discarded, missing
// future frames are pulled, and `out_mix` (may be NULL: advance only, map nothing) receives the
// neighbourhood of the timestamp, ready for pl_render_image_mix. It stays valid until the next
// pl_queue_update / pl_queue_reset. With PL_QUEUE_MORE the mix is still written but may be
// incomplete.
PL_API enum pl_queue_status pl_queue_update(pl_queue queue, struct pl_frame_mix *out_mix,
                                            const struct pl_queue_params *params);

// Estimated source frames / display refreshes per second (0 = unknown)
PL_API float pl_queue_estimate_fps(pl_queue queue);
PL_API float pl_queue_estimate_vps(pl_queue queue);

PL_API int pl_queue_num_frames(pl_queue queue);   // timeline entries currently held
PL_API double pl_queue_pts_offset(pl_queue queue); // what drift compensation adds to `pts`
PL_API bool pl_queue_peek(pl_queue queue, int idx, struct pl_source_frame *out);

PL_API_END

#endif // LIBPLACEBO_FRAME_QUEUE_H_
