/*
 * libplacebo-hip: pl_queue — a decoder's (frame, pts) stream as pl_frame_mix'es.
 *
 * Restates the behaviour of the reference's src/utils/frame_queue.c:
 *   rate estimation        :230-262     push / field pairing   :264-428
 *   back-pressure          :437-477     pull (get_frame)       :503-546
 *   lazy mapping           :548-590     advance / ZOH at EOF   :598-672
 *   point / oversample / interpolate :674-930    prefill :932-962    pl_queue_update :964-1054
 *
 * Layout differs from the reference: decoded *pictures* (user callbacks, textures, map state,
 * reference count) are separate objects from timeline *slots* (pts, signature, field, neighbour
 * pictures, the pl_frame handed out). A progressive frame is one slot on one picture, an
 * interlaced one two slots on one picture. Every externally visible result — status codes,
 * signatures, timestamps, which callbacks run and when — follows the reference, including its
 * float / double evaluation order; tests/test_frame_queue.py compares both on random traces.
 */
#include <errno.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <libplacebo/utils/frame_queue.h>

#include "host_common.h"

#define RATE_WINDOW       32    // samples averaged
#define RATE_WARMUP       4     // samples before a re-started average is trusted
#define RATE_JUMP         0.3f  // relative change that restarts the average
#define VSYNC_MIN_FPS     10    // display rates outside [10, 400] Hz are not believed
#define VSYNC_MAX_FPS     400
#define STICKY_RATIO      0.3   // hysteresis of `interpolation_threshold`
#define STICKY_FRAMES     5
#define UNMAPPED_AHEAD    2     // not yet mapped frames a blocking push lets through

struct rate {
    float ring[RATE_WINDOW];
    float sum, value;
    int pos, held, seen;
};

struct picture {
    int refs;
    struct pl_source_frame src;
    struct pl_frame frame;
    pl_tex tex[4];
    bool mapped, ok;
};

struct slot {
    double pts;
    uint64_t id;                     // signature in the mix
    struct picture *pic;
    bool second;                     // second field of `pic`
    enum pl_field field;
    struct picture *before, *after;  // temporal neighbours, for a deinterlacer
    bool unpaired;                   // first field still waiting for its second one
    bool exported;                   // handed out with neighbour references
    struct pl_frame view;
};

struct texset { pl_tex tex[4]; };

struct pl_queue_t {
    pl_gpu gpu;
    pl_log log;

    // `update_lock` serialises pl_queue_update / pl_queue_reset; `state_lock` guards the
    // fields below and is released while the user's get_frame runs, so pushes can go on
    pthread_mutex_t update_lock, state_lock;
    pthread_cond_t wake;

    struct slot **line;              // timeline, ascending pts
    int num, cap;
    uint64_t next_id;
    int sticky;
    bool starving, eof;

    struct rate fps, vps;            // seconds per frame / per vsync
    float told_fps, told_vps;
    double last_pts, pts_offset;

    // the mix handed out
    uint64_t *mix_id;
    float *mix_ts;
    const struct pl_frame **mix_frame;
    int mix_num, mix_cap;

    struct texset *spare;            // recycled texture sets
    int num_spare, cap_spare;
};

#define Q_MSG(q, lev, ...) pl_msg((q)->log, lev, __VA_ARGS__)

// Allocation failure is fatal, as in the reference's allocator (pl_alloc.c aborts on OOM)
static void *must(void *ptr)
{
    if (!ptr)
        abort();
    return ptr;
}

/* ---- rate estimation ---------------------------------------------------------------------- */

static inline float rel_change(float from, float to)
{
    return fabsf((to - from) / PL_MIN(to, from));
}

static void rate_seed(struct rate *r, float v)
{
    if (!r->value && isnormal(v) && v > 0.0)
        r->value = v;
}

static void rate_add(struct rate *r, float v)
{
    if (r->held && rel_change(r->sum / r->held, v) > RATE_JUMP) {
        r->sum = 0.0;
        r->held = r->pos = 0;
    }

    if (r->held == RATE_WINDOW) {
        r->sum -= r->ring[r->pos];
    } else {
        r->held++;
    }
    r->ring[r->pos] = v;
    r->sum += v;
    r->pos = (r->pos + 1) % RATE_WINDOW;
    r->seen++;

    if (r->seen < RATE_WARMUP || r->held >= RATE_WARMUP)
        r->value = r->sum / r->held;
}

static void tell_rates(pl_queue q)
{
    if (q->fps.seen < RATE_WARMUP || q->vps.seen < RATE_WARMUP)
        return;
    if (q->told_fps && q->told_vps && rel_change(q->told_fps, q->fps.value) < RATE_JUMP &&
        rel_change(q->told_vps, q->vps.value) < RATE_JUMP)
        return;
    Q_MSG(q, PL_LOG_INFO, "frame queue: source %.3f fps, display %.3f fps",
          1.0 / q->fps.value, 1.0 / q->vps.value);
    q->told_fps = q->fps.value;
    q->told_vps = q->vps.value;
}

/* ---- pictures and slots ------------------------------------------------------------------- */

static void picture_release(pl_queue q, struct picture **ppic, bool recycle)
{
    struct picture *pic = *ppic;
    *ppic = NULL;
    if (!pic || --pic->refs)
        return;

    if (!pic->mapped && pic->src.discard)
        pic->src.discard(&pic->src);
    if (pic->mapped && pic->ok && pic->src.unmap)
        pic->src.unmap(q->gpu, &pic->frame, &pic->src);

    bool any = false;
    for (int i = 0; i < 4; i++) {
        if (!pic->tex[i])
            continue;
        any = true;
        if (recycle) {
            pl_tex_invalidate(q->gpu, pic->tex[i]);
        } else {
            pl_tex_destroy(q->gpu, &pic->tex[i]);
        }
    }
    if (recycle && any) {
        if (q->num_spare == q->cap_spare) {
            q->cap_spare = PL_MAX(8, 2 * q->cap_spare);
            q->spare = must(realloc(q->spare, q->cap_spare * sizeof(*q->spare)));
        }
        memcpy(q->spare[q->num_spare++].tex, pic->tex, sizeof(pic->tex));
    }
    free(pic);
}

static struct picture *picture_ref(struct picture *pic)
{
    pic->refs++;
    return pic;
}

// A slot leaving the timeline lets go of its pictures at once (the neighbour references of
// different slots would otherwise keep each other alive)
static void slot_free(pl_queue q, struct slot *s, bool recycle)
{
    if (s->second)
        picture_release(q, &s->pic, recycle);
    picture_release(q, &s->before, recycle);
    picture_release(q, &s->after, recycle);
    picture_release(q, &s->pic, recycle);
    free(s);
}

static void line_insert(pl_queue q, int at, struct slot *s)
{
    if (q->num == q->cap) {
        q->cap = PL_MAX(16, 2 * q->cap);
        q->line = must(realloc(q->line, q->cap * sizeof(*q->line)));
    }
    memmove(&q->line[at + 1], &q->line[at], (q->num - at) * sizeof(*q->line));
    q->line[at] = s;
    q->num++;
}

static struct slot *slot_new(pl_queue q, struct picture *pic, double pts, bool second)
{
    struct slot *s = must(calloc(1, sizeof(*s)));
    s->pic = picture_ref(pic);
    s->pts = pts;
    s->second = second;
    return s;
}

/* ---- lifetime ------------------------------------------------------------------------------ */

pl_queue pl_queue_create(pl_gpu gpu)
{
    pl_queue q = calloc(1, sizeof(*q));
    if (!q)
        return NULL;
    q->gpu = gpu;
    q->log = gpu->log;
    pthread_mutex_init(&q->update_lock, NULL);
    pthread_mutex_init(&q->state_lock, NULL);

    pthread_condattr_t attr;
    pthread_condattr_init(&attr);
    pthread_condattr_setclock(&attr, CLOCK_MONOTONIC);
    const int err = pthread_cond_init(&q->wake, &attr);
    pthread_condattr_destroy(&attr);
    if (err) {
        Q_MSG(q, PL_LOG_ERR, "frame queue: pthread_cond_init failed (%d)", err);
        pthread_mutex_destroy(&q->update_lock);
        pthread_mutex_destroy(&q->state_lock);
        free(q);
        return NULL;
    }
    return q;
}

static void drop_all(pl_queue q)
{
    for (int i = 0; i < q->num; i++)
        slot_free(q, q->line[i], false);
    q->num = 0;
}

void pl_queue_destroy(pl_queue *queue)
{
    pl_queue q = *queue;
    if (!q)
        return;
    drop_all(q);
    for (int n = 0; n < q->num_spare; n++) {
        for (int i = 0; i < 4; i++)
            pl_tex_destroy(q->gpu, &q->spare[n].tex[i]);
    }
    pthread_cond_destroy(&q->wake);
    pthread_mutex_destroy(&q->state_lock);
    pthread_mutex_destroy(&q->update_lock);
    free(q->line); free(q->mix_id); free(q->mix_ts); free(q->mix_frame); free(q->spare);
    free(q);
    *queue = NULL;
}

void pl_queue_reset(pl_queue q)
{
    pthread_mutex_lock(&q->update_lock);
    pthread_mutex_lock(&q->state_lock);

    drop_all(q);
    q->next_id = 0;
    q->sticky = 0;
    q->starving = q->eof = false;
    memset(&q->fps, 0, sizeof(q->fps));
    memset(&q->vps, 0, sizeof(q->vps));
    q->told_fps = q->told_vps = 0;
    q->last_pts = q->pts_offset = 0;
    q->mix_num = 0;

    pthread_cond_signal(&q->wake);
    pthread_mutex_unlock(&q->state_lock);
    pthread_mutex_unlock(&q->update_lock);
}

// pthread_cond_timedwait on a relative timeout in ns (UINT64_MAX = no limit)
static int wait_wake(pl_queue q, uint64_t timeout)
{
    if (timeout == UINT64_MAX)
        return pthread_cond_wait(&q->wake, &q->state_lock);
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    ts.tv_sec += timeout / 1000000000ull;
    ts.tv_nsec += timeout % 1000000000ull;
    if (ts.tv_nsec >= 1000000000l) {
        ts.tv_nsec -= 1000000000l;
        ts.tv_sec++;
    }
    return pthread_cond_timedwait(&q->wake, &q->state_lock, &ts);
}

/* ---- pushing ------------------------------------------------------------------------------- */

// Is `s` the slot `anchor` or, when `anchor` is a second field, the first field next to it?
static inline bool of_anchor(const struct slot *s, const struct slot *anchor)
{
    return s == anchor || (anchor->second && !s->second && s->pic == anchor->pic);
}

// Point a neighbour link of an already queued slot at the new picture. A slot that was
// already shown with its old link gets a fresh signature, so cached renders of it are redone.
static void relink(pl_queue q, struct slot *s, struct picture **link, struct picture *to)
{
    picture_release(q, link, true);
    *link = picture_ref(to);
    if (s->exported) {
        s->id = q->next_id++;
        s->exported = false;
    }
}

static void push_locked(pl_queue q, const struct pl_source_frame *src)
{
    if (q->eof) {
        if (src) {
            Q_MSG(q, PL_LOG_INFO, "frame queue: frame pushed after EOF, dropping it");
            if (src->discard)
                src->discard(src);
        }
        return;
    }

    pthread_cond_signal(&q->wake);
    if (!src) {
        q->eof = true;
        q->starving = false;
        if (q->num && q->line[q->num - 1]->unpaired) {
            Q_MSG(q, PL_LOG_WARN, "frame queue: EOF before the second field of the frame at "
                  "%f could be placed; it is dropped", q->line[q->num - 1]->pts);
        }
        return;
    }

    // frame rate: the stated duration seeds it, pts differences refine it
    rate_seed(&q->fps, src->first_field ? src->duration / 2 : src->duration);
    if (q->num) {
        const double last = q->line[q->num - 1]->pts;
        const float step = src->pts - last;
        if (step <= 0.0f) {
            Q_MSG(q, PL_LOG_DEBUG, "frame queue: pts went back, %f -> %f", last, src->pts);
        } else if (q->fps.value && step > 10.0 * q->fps.value) {
            Q_MSG(q, PL_LOG_DEBUG, "frame queue: pts jumped, %f -> %f", last, src->pts);
        } else {
            rate_add(&q->fps, step);
        }
    }

    struct picture *pic = must(calloc(1, sizeof(*pic)));
    pic->src = *src;
    struct slot *first = slot_new(q, pic, src->pts, false);
    first->id = q->next_id++;
    if (q->num_spare)
        memcpy(pic->tex, q->spare[--q->num_spare].tex, sizeof(pic->tex));

    int at = q->num; // after the last slot that is not later
    while (at > 0 && q->line[at - 1]->pts > first->pts)
        at--;

    q->starving = false;
    if (src->first_field == PL_FIELD_NONE) {
        line_insert(q, at, first);
        return;
    }

    struct slot *prev = at > 0 ? q->line[at - 1] : NULL;
    struct slot *next = at < q->num ? q->line[at] : NULL;
    if (prev && prev->unpaired) {
        // now there is a time for the second field of the frame before this one
        struct slot *pair = slot_new(q, prev->pic, (prev->pts + first->pts) / 2, true);
        pair->field = pl_field_other(prev->field);
        pair->id = q->next_id++;
        line_insert(q, at++, pair);
        prev->unpaired = false;
        prev = pair;
    }

    first->field = src->first_field;
    double pair_pts;
    if (next) {
        pair_pts = (first->pts + next->pts) / 2;
    } else if (src->duration) {
        pair_pts = first->pts + src->duration / 2;
    } else if (q->fps.value) {
        pair_pts = first->pts + q->fps.value;
    } else {
        Q_MSG(q, PL_LOG_DEBUG, "frame queue: interlaced frame at %f, frame rate unknown; "
              "second field deferred", src->pts);
        first->unpaired = true;
        line_insert(q, at, first);
        return;
    }

    struct slot *second = slot_new(q, pic, pair_pts, true);
    second->field = pl_field_other(first->field);
    second->id = q->next_id++;

    if (prev) {
        first->before = picture_ref(prev->pic);
        second->before = picture_ref(prev->pic);
        for (int j = at - 1; j >= 0 && of_anchor(q->line[j], prev); j--)
            relink(q, q->line[j], &q->line[j]->after, pic);
    }
    if (next) {
        first->after = picture_ref(next->pic);
        second->after = picture_ref(next->pic);
        for (int j = at; j < q->num && of_anchor(q->line[j], next); j++)
            relink(q, q->line[j], &q->line[j]->before, pic);
    }

    line_insert(q, at, first);
    line_insert(q, at + 1, second);
}

void pl_queue_push(pl_queue q, const struct pl_source_frame *frame)
{
    pthread_mutex_lock(&q->state_lock);
    push_locked(q, frame);
    pthread_mutex_unlock(&q->state_lock);
}

// Back-pressure for decoder threads: room while the consumer is waiting, while the tail is in
// use, or while fewer unmapped frames wait than one vsync can consume (+ UNMAPPED_AHEAD - 1)
static bool has_room(pl_queue q)
{
    if (q->starving)
        return true;

    int allowed = UNMAPPED_AHEAD;
    if (q->fps.value && q->vps.value && q->vps.value <= 1.0f / VSYNC_MIN_FPS)
        allowed += ceilf(q->vps.value / q->fps.value) - 1;

    for (int i = q->num - 1; i >= 0; i--) {
        if (q->line[i]->pic->mapped)
            return true;
        if (q->num - i >= allowed)
            return false;
    }
    return true;
}

bool pl_queue_push_block(pl_queue q, uint64_t timeout, const struct pl_source_frame *frame)
{
    pthread_mutex_lock(&q->state_lock);
    if (timeout && frame) {
        while (!q->eof && !has_room(q)) {
            if (wait_wake(q, timeout) == ETIMEDOUT) {
                pthread_mutex_unlock(&q->state_lock);
                return false;
            }
        }
    }
    push_locked(q, frame);
    pthread_mutex_unlock(&q->state_lock);
    return true;
}

/* ---- pulling and mapping ------------------------------------------------------------------- */

// One more frame, from the callback or from a pusher. Called with state_lock held; the lock is
// released while waiting, so more than one frame may have arrived on return.
static enum pl_queue_status pull(pl_queue q, const struct pl_queue_params *params)
{
    if (q->eof)
        return PL_QUEUE_EOF;

    if (params->get_frame) {
        pthread_mutex_unlock(&q->state_lock);
        struct pl_source_frame src;
        const enum pl_queue_status st = params->get_frame(&src, params);
        if (st == PL_QUEUE_OK) {
            pl_queue_push(q, &src);
        } else if (st == PL_QUEUE_EOF) {
            pl_queue_push(q, NULL);
        }
        pthread_mutex_lock(&q->state_lock);
        return st;
    }

    if (!params->timeout)
        return PL_QUEUE_MORE;
    q->starving = true;
    pthread_cond_signal(&q->wake);
    while (q->starving) {
        if (wait_wake(q, params->timeout) == ETIMEDOUT)
            return PL_QUEUE_MORE;
    }
    return q->eof ? PL_QUEUE_EOF : PL_QUEUE_OK;
}

static bool picture_map(pl_queue q, struct picture *pic)
{
    if (!pic->mapped) {
        pic->mapped = true;
        pic->ok = pic->src.map(q->gpu, pic->tex, &pic->src, &pic->frame);
        if (!pic->ok)
            Q_MSG(q, PL_LOG_ERR, "frame queue: mapping the frame at %f failed", pic->src.pts);
    }
    return pic->ok;
}

// Map what showing `s` needs and (re)build the pl_frame handed out for it
static bool slot_map(pl_queue q, struct slot *s)
{
    bool ok = picture_map(q, s->pic);
    if (s->before)
        ok &= picture_map(q, s->before);
    if (s->after)
        ok &= picture_map(q, s->after);
    if (!ok)
        return false;

    s->view = s->pic->frame;
    if (s->field) {
        s->view.field = s->field;
        s->view.first_field = s->pic->src.first_field;
        s->view.prev = s->before ? &s->before->frame : NULL;
        s->view.next = s->after ? &s->after->frame : NULL;
        s->exported = true;
    }
    return true;
}

// A field is complete once the following picture is known (or will never come)
static inline bool slot_complete(pl_queue q, const struct slot *s)
{
    return !s->field || s->after || q->eof;
}

/* ---- moving along the timeline ------------------------------------------------------------- */

// Make line[0] the last slot at or before `pts` and pull until a later one exists. After EOF
// the last slot is held for one more frame duration, then the queue reports EOF.
static enum pl_queue_status advance(pl_queue q, double pts, const struct pl_queue_params *params)
{
    enum pl_queue_status st;
    for (;;) {
        int gone = 0;
        while (gone + 1 < q->num && q->line[gone + 1]->pts <= pts)
            slot_free(q, q->line[gone++], true);
        if (gone) {
            memmove(q->line, q->line + gone, (q->num - gone) * sizeof(*q->line));
            q->num -= gone;
        }

        st = PL_QUEUE_OK;
        if (q->num && q->line[q->num - 1]->pts > pts)
            break;
        st = pull(q, params);
        if (st == PL_QUEUE_ERR || st == PL_QUEUE_MORE)
            return st;
        if (st == PL_QUEUE_EOF) {
            if (!q->num)
                return st;
            goto tail;
        }
    }

    if (!slot_complete(q, q->line[PL_MIN(q->num - 1, 1)])) {
        const enum pl_queue_status more = pull(q, params);
        if (more == PL_QUEUE_ERR)
            return more;
        if (more == PL_QUEUE_MORE)
            st = PL_QUEUE_MORE;
    }

tail:
    if (q->eof && q->num == 1) {
        // a lone frame at pts 0, or no frame rate at all: a still image, shown for ever
        if (q->line[0]->pts == 0.0 || !q->fps.value)
            return PL_QUEUE_OK;
        if (pts < q->line[0]->pts + q->fps.value)
            return PL_QUEUE_OK;
        slot_free(q, q->line[0], true);
        q->num = 0;
        return PL_QUEUE_EOF;
    }
    return st;
}

static void mix_clear(pl_queue q)
{
    q->mix_num = 0;
}

static void mix_add(pl_queue q, const struct slot *s, float ts)
{
    if (q->mix_num == q->mix_cap) {
        q->mix_cap = PL_MAX(16, 2 * q->mix_cap);
        q->mix_id = must(realloc(q->mix_id, q->mix_cap * sizeof(*q->mix_id)));
        q->mix_ts = must(realloc(q->mix_ts, q->mix_cap * sizeof(*q->mix_ts)));
        q->mix_frame = must(realloc(q->mix_frame, q->mix_cap * sizeof(*q->mix_frame)));
    }
    q->mix_id[q->mix_num] = s->id;
    q->mix_ts[q->mix_num] = ts;
    q->mix_frame[q->mix_num++] = &s->view;
}

static void mix_out(pl_queue q, struct pl_frame_mix *mix, float vsync_duration)
{
    *mix = (struct pl_frame_mix) {
        .num_frames     = q->mix_num,
        .frames         = q->mix_frame,
        .signatures     = q->mix_id,
        .timestamps     = q->mix_ts,
        .vsync_duration = vsync_duration,
    };
}

// The one frame nearest to the timestamp (nothing before the first frame is due)
static enum pl_queue_status pick_nearest(pl_queue q, struct pl_frame_mix *mix,
                                         const struct pl_queue_params *params)
{
    *mix = (struct pl_frame_mix) {0};
    if (!q->num)
        return PL_QUEUE_MORE;
    if (q->line[0]->pts > params->pts)
        return PL_QUEUE_OK;

    int best = 0;
    double dist = fabs(q->line[0]->pts - params->pts);
    for (int i = 1; i < q->num; i++) {
        const double d = fabs(q->line[i]->pts - params->pts);
        if (!(d < dist))
            break;
        best = i;
        dist = d;
    }

    struct slot *s = q->line[best];
    if (!slot_map(q, s))
        return PL_QUEUE_ERR;
    mix_clear(q);
    mix_add(q, s, 0.0);
    mix_out(q, mix, 1.0);
    tell_rates(q);
    return slot_complete(q, s) ? PL_QUEUE_OK : PL_QUEUE_MORE;
}

static enum pl_queue_status show_nearest(pl_queue q, struct pl_frame_mix *mix,
                                         const struct pl_queue_params *params)
{
    const enum pl_queue_status st = advance(q, params->pts, params);
    if (st == PL_QUEUE_ERR || st == PL_QUEUE_EOF)
        return st;
    if (mix && pick_nearest(q, mix, params) == PL_QUEUE_ERR)
        return PL_QUEUE_ERR;
    return st;
}

// Mixer without a radius ("oversample"): exactly the frames either side of the timestamp
static enum pl_queue_status show_pair(pl_queue q, struct pl_frame_mix *mix,
                                      const struct pl_queue_params *params)
{
    const enum pl_queue_status st = advance(q, params->pts, params);
    if (st == PL_QUEUE_ERR || st == PL_QUEUE_EOF)
        return st;
    if (st == PL_QUEUE_MORE && !q->num) {
        if (mix)
            *mix = (struct pl_frame_mix) {0};
        return st;
    }
    if (!mix)
        return PL_QUEUE_OK;

    if (q->num < 2 || q->line[0]->pts > params->pts) {
        if (pick_nearest(q, mix, params) == PL_QUEUE_ERR)
            return PL_QUEUE_ERR;
        return st;
    }

    mix_clear(q);
    for (int i = 0; i < 2; i++) {
        struct slot *s = q->line[i];
        if (!slot_map(q, s))
            return PL_QUEUE_ERR;
        mix_add(q, s, (s->pts - params->pts) / q->fps.value);
    }
    mix_out(q, mix, q->vps.value / q->fps.value);
    tell_rates(q);
    return st;
}

// Everything within the mixer's radius of the timestamp, in units of source frames
static enum pl_queue_status show_mix(pl_queue q, struct pl_frame_mix *mix,
                                     const struct pl_queue_params *params)
{
    if (!q->fps.value)
        return show_nearest(q, mix, params); // a still image, or the very first frame

    // display and source rate (nearly) equal: single frames, with some hysteresis
    const float ratio = fabs(q->fps.value / q->vps.value - 1.0);
    if (ratio <= params->interpolation_threshold) {
        if (!q->sticky) {
            Q_MSG(q, PL_LOG_INFO, "frame queue: rate ratio %.4f within the threshold %.4f, "
                  "not interpolating", ratio, params->interpolation_threshold);
        }
        q->sticky = STICKY_FRAMES + 1;
        return show_nearest(q, mix, params);
    } else if (ratio < STICKY_RATIO && q->sticky > 1) {
        q->sticky--;
        return show_nearest(q, mix, params);
    }
    if (q->sticky) {
        Q_MSG(q, PL_LOG_INFO, "frame queue: rate ratio %.4f beyond the threshold %.4f, "
              "interpolating again", ratio, params->interpolation_threshold);
    }
    q->sticky = 0;

    if (!params->radius)
        return show_pair(q, mix, params);

    const float radius = params->radius * fmaxf(1.0f, q->vps.value / q->fps.value);
    const double from = params->pts - radius * q->fps.value,
                 upto = params->pts + radius * q->fps.value;

    enum pl_queue_status st = advance(q, from, params);
    if (st == PL_QUEUE_ERR || st == PL_QUEUE_EOF)
        return st;
    if (st == PL_QUEUE_OK) {
        if (q->line[0]->pts > params->pts)
            return show_nearest(q, mix, params); // the first frame is not due yet

        bool covered = true;
        while (q->line[q->num - 1]->pts < upto) {
            st = pull(q, params);
            if (st == PL_QUEUE_ERR)
                return st;
            if (st == PL_QUEUE_EOF) {
                // the last frame stays up for its own duration before EOF is passed on
                const double last = q->line[q->num - 1]->pts;
                if (last && params->pts >= last + q->fps.value)
                    return st;
                st = PL_QUEUE_OK;
                covered = false;
                break;
            }
            if (st == PL_QUEUE_MORE) {
                covered = false;
                break;
            }
        }

        if (covered) {
            int last = q->num - 1;
            while (last && q->line[last]->pts > upto)
                last--;
            if (!slot_complete(q, q->line[last])) {
                st = pull(q, params);
                if (st == PL_QUEUE_ERR || st == PL_QUEUE_EOF)
                    return st;
            }
        }
    }

    if (!mix)
        return PL_QUEUE_OK;

    // from the last frame before the window (the zero-order-hold fallback) to its end
    mix_clear(q);
    for (int i = 0; i < q->num && q->line[i]->pts <= upto; i++) {
        struct slot *s = q->line[i];
        if (!slot_map(q, s))
            return PL_QUEUE_ERR;
        mix_add(q, s, (s->pts - params->pts) / q->fps.value);
    }
    mix_out(q, mix, q->vps.value / q->fps.value);
    tell_rates(q);
    return st;
}

// First update of a fresh queue: gather what the mixer will want and map it right away, so the
// GPU-side set-up happens before timing matters
static bool prefill(pl_queue q, const struct pl_queue_params *params)
{
    int want = 2 * ceilf(params->radius);
    if (q->fps.value && q->vps.value && q->vps.value <= 1.0f / VSYNC_MIN_FPS)
        want *= ceilf(q->vps.value / q->fps.value);
    want = PL_MAX(want, UNMAPPED_AHEAD);

    while (q->num < want) {
        const enum pl_queue_status st = pull(q, params);
        if (st == PL_QUEUE_ERR)
            return false;
        if (st != PL_QUEUE_OK)
            return true;
    }
    for (int i = 0; i < want; i++) {
        if (!slot_map(q, q->line[i]))
            return false;
    }
    return true;
}

enum pl_queue_status pl_queue_update(pl_queue q, struct pl_frame_mix *out_mix,
                                     const struct pl_queue_params *params)
{
    enum pl_queue_status st = PL_QUEUE_ERR;
    struct pl_queue_params snapped;
    pthread_mutex_lock(&q->update_lock);
    pthread_mutex_lock(&q->state_lock);
    rate_seed(&q->vps, params->vsync_duration);

    const float step = params->pts - q->last_pts;
    if (step < 0.0f) {
        // going back is fine as long as the frame for that time is still there
        if (q->num && q->line[0]->pts > params->pts) {
            Q_MSG(q, PL_LOG_ERR, "frame queue: pts %f requested, but the oldest frame left is at "
                  "%f. Timestamps must not decrease; use pl_queue_reset to seek.",
                  params->pts, q->line[0]->pts);
            goto done;
        }
    } else if (step > 1.0f) {
        q->pts_offset = 0.0; // resumed after a pause: not a vsync interval
    } else if (step > 0) {
        rate_add(&q->vps, params->pts - q->last_pts);
    }
    q->last_pts = params->pts;

    if (params->drift_compensation > 0.0f) {
        // a timestamp that almost hits a frame is meant to hit it; keep the difference
        double pts = params->pts + q->pts_offset;
        for (int i = 0; i < q->num; i++) {
            if (fabs(q->line[i]->pts - pts) < params->drift_compensation) {
                q->pts_offset = q->line[i]->pts - params->pts;
                pts = q->line[i]->pts;
                break;
            }
        }
        snapped = *params;
        snapped.pts = pts;
        params = &snapped;
    }

    if (!params->pts && !q->num && !prefill(q, params))
        goto done;

    static const float vsync_longest = 1.0 / VSYNC_MIN_FPS, vsync_shortest = 1.0 / VSYNC_MAX_FPS;
    const bool vps_known = q->vps.value > vsync_shortest && q->vps.value < vsync_longest;
    if (vps_known || params->vsync_duration > 0) {
        st = show_mix(q, out_mix, params);
    } else {
        st = show_nearest(q, out_mix, params);
    }
    if (st == PL_QUEUE_ERR)
        Q_MSG(q, PL_LOG_ERR, "frame queue: update to pts %f failed", params->pts);

done:
    pthread_cond_signal(&q->wake);
    pthread_mutex_unlock(&q->state_lock);
    pthread_mutex_unlock(&q->update_lock);
    return st;
}

/* ---- queries -------------------------------------------------------------------------------- */

float pl_queue_estimate_fps(pl_queue q)
{
    pthread_mutex_lock(&q->state_lock);
    const float period = q->fps.value;
    pthread_mutex_unlock(&q->state_lock);
    return period ? 1.0f / period : 0.0f;
}

float pl_queue_estimate_vps(pl_queue q)
{
    pthread_mutex_lock(&q->state_lock);
    const float period = q->vps.value;
    pthread_mutex_unlock(&q->state_lock);
    return period ? 1.0f / period : 0.0f;
}

int pl_queue_num_frames(pl_queue q)
{
    pthread_mutex_lock(&q->state_lock);
    const int num = q->num;
    pthread_mutex_unlock(&q->state_lock);
    return num;
}

double pl_queue_pts_offset(pl_queue q)
{
    pthread_mutex_lock(&q->state_lock);
    const double offset = q->pts_offset;
    pthread_mutex_unlock(&q->state_lock);
    return offset;
}

bool pl_queue_peek(pl_queue q, int idx, struct pl_source_frame *out)
{
    pthread_mutex_lock(&q->state_lock);
    const bool ok = idx >= 0 && idx < q->num;
    if (ok) {
        // like the reference, a second-field entry has no source frame of its own
        if (q->line[idx]->second) {
            memset(out, 0, sizeof(*out));
        } else {
            *out = q->line[idx]->pic->src;
        }
    }
    pthread_mutex_unlock(&q->state_lock);
    return ok;
}
