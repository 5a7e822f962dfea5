/*
 * libplacebo-hip — the one cross-GPU exchange of the render path: all-reduce of the HDR peak
 * measurement over RCCL (include/libplacebo/hip.h). Everything else about
 * multi-GPU operation is "one pl_hip + pl_renderer per GPU", i.e. no code.
 *
 * RCCL is bound at run time: the library must load (and the CPU test suite must run) on hosts
 * without librccl.so.
 */
#include <dlfcn.h>
#include <stdlib.h>

#include <libplacebo/hip.h>

#include "gpu_priv.h"

enum { NCCL_UINT32 = 3, NCCL_SUM = 0, NCCL_MAX = 2 };   // rccl.h:448-462
enum { PEAK_WORDS = 816, MAX_FIRST = 36, MAX_COUNT = 12 };

typedef int (*allreduce_fn)(const void *send, void *recv, size_t count, int dtype, int op,
                            void *comm, void *stream);

struct pl_hip_rccl_t {
    pl_gpu gpu;
    void *comm;
    void *dl;
    allreduce_fn all_reduce;
    uint32_t *maxima;       // device scratch: the MAX words travel separately
    uint32_t *summed;       // device scratch: the SUM arrives here, not over the local measurement
    int exchanges, errors;
};

void pl_hip_set_peak_exchange(pl_gpu gpu, pl_hip_peak_exchange_fn fn, void *priv)
{
    struct gpu_priv *p = GPU_PRIV(gpu);
    p->peak_exchange = fn;
    p->peak_exchange_priv = priv;
}

void plh_gpu_peak_exchange(pl_gpu gpu, void *words, size_t size)
{
    struct gpu_priv *p = GPU_PRIV(gpu);
    if (p->peak_exchange)
        p->peak_exchange(p->peak_exchange_priv, words, size, p->stream);
}

pl_hip_rccl pl_hip_rccl_create(pl_gpu gpu, void *nccl_comm, void *nccl_all_reduce)
{
    if (!nccl_comm) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_hip_rccl_create: NULL communicator");
        return NULL;
    }
    struct pl_hip_rccl_t *x = calloc(1, sizeof(*x));
    if (!x)
        return NULL;
    x->gpu = gpu;
    x->comm = nccl_comm;
    static const char *const names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so" };
    x->all_reduce = (allreduce_fn) nccl_all_reduce;
    if (!x->all_reduce)
        x->all_reduce = (allreduce_fn) dlsym(RTLD_DEFAULT, "ncclAllReduce");
    for (size_t i = 0; !x->all_reduce && i < PL_ARRAY_SIZE(names); i++) {
        x->dl = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (x->dl)
            x->all_reduce = (allreduce_fn) dlsym(x->dl, "ncclAllReduce");
    }
    if (!x->all_reduce) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_hip_rccl_create: ncclAllReduce not found (no librccl)");
        pl_hip_rccl_destroy(&x);
        return NULL;
    }
    x->maxima = plh_malloc(plh_gpu_device(gpu), MAX_COUNT * sizeof(uint32_t));
    x->summed = plh_malloc(plh_gpu_device(gpu), PEAK_WORDS * sizeof(uint32_t));
    if (!x->maxima || !x->summed)
        pl_hip_rccl_destroy(&x);
    return x;
}

void pl_hip_rccl_destroy(pl_hip_rccl *px)
{
    struct pl_hip_rccl_t *x = px ? *px : NULL;
    if (!x)
        return;
    if (x->maxima || x->summed) {
        plh_stream_sync(plh_gpu_stream(x->gpu));
        plh_free(x->maxima);
        plh_free(x->summed);
    }
    if (x->dl)
        dlclose(x->dl);
    free(x);
    *px = NULL;
}

int pl_hip_rccl_stats(pl_hip_rccl x, int *out_errors)
{
    if (out_errors)
        *out_errors = x->errors;
    return x->exchanges;
}

// SUM over everything, MAX over the frame_max_pq words: the maxima are set aside and reduced with
// MAX, the whole buffer is summed OUT OF PLACE, and only when every step was accepted do the two
// results replace the local measurement (the sum first -- its max words are garbage -- then the
// maxima over them). A refused collective therefore leaves `words` exactly as the measuring pass
// wrote it: the renderer tone-maps with this GPU's own measurement instead of a half-reduced
// buffer. Four tiny stream-ordered copies + two latency-bound collectives (3.3 KB / 48 B).
void pl_hip_rccl_peak_exchange(void *priv, void *words, size_t size, void *stream)
{
    struct pl_hip_rccl_t *x = priv;
    if (size != PEAK_WORDS * sizeof(uint32_t)) {
        x->errors++;
        return;
    }
    uint32_t *w = words;
    const size_t mbytes = MAX_COUNT * sizeof(uint32_t), bytes = PEAK_WORDS * sizeof(uint32_t);
    x->exchanges++;
    int rc = plh_copy2d_d2d(stream, x->maxima, mbytes, w + MAX_FIRST, mbytes, mbytes, 1);
    if (!rc)
        rc = x->all_reduce(w, x->summed, PEAK_WORDS, NCCL_UINT32, NCCL_SUM, x->comm, stream);
    if (!rc)
        rc = x->all_reduce(x->maxima, x->maxima, MAX_COUNT, NCCL_UINT32, NCCL_MAX, x->comm, stream);
    if (rc) {
        x->errors++;
        pl_msg(x->gpu->log, PL_LOG_ERR, "peak exchange over RCCL failed (status %d): this frame is "
               "tone-mapped with the local measurement", rc);
        return;
    }
    rc = plh_copy2d_d2d(stream, w, bytes, x->summed, bytes, bytes, 1);
    rc |= plh_copy2d_d2d(stream, w + MAX_FIRST, mbytes, x->maxima, mbytes, mbytes, 1);
    if (rc) {
        x->errors++;
        pl_msg(x->gpu->log, PL_LOG_ERR, "peak exchange: copying the reduced measurement back failed (%d)", rc);
    }
}
