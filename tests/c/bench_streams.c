/*
 * The multi-GPU shape of SURVEY.md 8(e) in one process, from plain C: N host threads, each with
 * its own pl_hip + pl_renderer on device (thread % devices), rendering independent streams of the
 * metric's frame (HDR10 1080p -> EWA-Lanczos 2x -> tone + gamut map -> BT.1886, 10-bit dither,
 * 4K). No data-path collective: streams share nothing. With --scene-peak the streams render
 * frames of ONE scene: the 816-word peak measurement is all-reduced over RCCL
 * (ncclCommInitAll, one communicator per device; pl_hip_rccl_*, include/libplacebo/hip.h) before
 * every tone curve -- the only exchange of the path.
 *
 * usage: bench_streams [streams] [frames] [--scene-peak]
 *        streams = 0: one per visible device (default). More streams than devices is allowed
 *        (several pl_hip on one device are independent), except with --scene-peak, where a
 *        communicator has one rank per device.
 * prints ONE JSON line: aggregate output Mpixels/s = all frames of all streams / wall time from
 * the common start (after a barrier) to the last stream's pl_gpu_finish.
 */
#include <dlfcn.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <libplacebo/hip.h>
#include <libplacebo/renderer.h>
#include <libplacebo/shaders/dithering.h>

#define SW 1920
#define SH 1080
#define DW 3840
#define DH 2160
#define POOL 6
#define WARMUP 30

static double now_us(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

struct stream {
    int index, device, frames;
    pl_log log;
    const uint16_t *pixels;
    pthread_barrier_t *start;
    void *nccl_comm, *nccl_all_reduce;   // --scene-peak
    double t_begin, t_end;               // wall clock of this stream's timed region
    int errors, exchanges, exchange_errors;
    const char *failure;
};

static void *stream_main(void *arg)
{
    struct stream *st = arg;
    pl_hip hip = pl_hip_create(st->log, pl_hip_params(.device = st->device));
    pl_hip_rccl rccl = NULL;
    pl_renderer rr = NULL;
    pl_tex src[POOL] = {0}, dst[POOL] = {0};
    if (!hip) {
        st->failure = "pl_hip_create failed";
        pthread_barrier_wait(st->start);
        return NULL;
    }
    pl_gpu gpu = hip->gpu;
    pl_fmt fmt = pl_find_named_fmt(gpu, "rgba16");
    for (int i = 0; i < POOL && !st->failure; i++) {
        src[i] = pl_tex_create(gpu, pl_tex_params(.w = SW, .h = SH, .format = fmt, .sampleable = true,
                                                  .host_writable = true, .initial_data = st->pixels));
        dst[i] = pl_tex_create(gpu, pl_tex_params(.w = DW, .h = DH, .format = fmt, .renderable = true,
                                                  .storable = true, .host_readable = true));
        if (!src[i] || !dst[i])
            st->failure = "texture creation failed";
    }
    if (st->nccl_comm && !st->failure) {
        rccl = pl_hip_rccl_create(gpu, st->nccl_comm, st->nccl_all_reduce);
        if (rccl)
            pl_hip_set_peak_exchange(gpu, pl_hip_rccl_peak_exchange, rccl);
        else
            st->failure = "pl_hip_rccl_create failed";
    }

    const struct pl_color_repr full16 = {
        .sys = PL_COLOR_SYSTEM_RGB, .levels = PL_COLOR_LEVELS_FULL,
        .bits = { .sample_depth = 16, .color_depth = 16 },
    };
    struct pl_color_repr ten_bit = full16;
    ten_bit.bits.color_depth = 10;
    ten_bit.bits.bit_shift = 6;
    struct pl_color_space hdr10 = pl_color_space_hdr10;
    hdr10.hdr.max_luma = 1000.0f;
    const struct pl_color_space bt1886 = { .primaries = PL_COLOR_PRIM_BT_709, .transfer = PL_COLOR_TRC_BT_1886 };
    struct pl_dither_params dither = pl_dither_default_params;
    struct pl_peak_detect_params peak = pl_peak_detect_default_params;
    peak.percentile = 99.995f;
    struct pl_render_params params = pl_render_default_params;     // bench.py: ewa_1080p_to_4k_hdr_tonemap
    params.upscaler = &pl_filter_ewa_lanczos;
    params.dither_params = &dither;
    params.peak_detect_params = &peak;

    struct pl_frame image = {
        .num_planes = 1,
        .planes = {{ .components = 3, .component_mapping = {0, 1, 2} }},
        .repr = full16, .color = hdr10,
    };
    struct pl_frame target = {
        .num_planes = 1,
        .planes = {{ .components = 4, .component_mapping = {0, 1, 2, 3} }},
        .repr = ten_bit, .color = bt1886,
    };
    if (!st->failure)
        rr = pl_renderer_create(st->log, gpu);
    // every stream renders the same number of frames (warm-up included): with --scene-peak each
    // frame is a collective, and a stream that stopped early would leave the others waiting
    for (int f = -WARMUP; f < st->frames; f++) {
        if (f == 0) {
            if (!st->failure)
                pl_gpu_finish(gpu);
            pthread_barrier_wait(st->start);
            st->t_begin = now_us();
        }
        if (st->failure)
            continue;
        image.planes[0].texture = src[(f + WARMUP) % POOL];
        target.planes[0].texture = dst[(f + WARMUP) % POOL];
        if (!pl_render_image(rr, &image, &target, &params))
            st->failure = "pl_render_image failed";
    }
    if (!st->failure)
        pl_gpu_finish(gpu);
    st->t_end = now_us();
    if (rr)
        st->errors = pl_renderer_get_errors(rr).errors;
    if (rccl)
        st->exchanges = pl_hip_rccl_stats(rccl, &st->exchange_errors);

    pl_renderer_destroy(&rr);
    if (rccl) {
        pl_hip_set_peak_exchange(gpu, NULL, NULL);
        pl_hip_rccl_destroy(&rccl);
    }
    for (int i = 0; i < POOL; i++) {
        pl_tex_destroy(gpu, &src[i]);
        pl_tex_destroy(gpu, &dst[i]);
    }
    pl_hip_destroy(&hip);
    return NULL;
}

int main(int argc, char **argv)
{
    int streams = 0, frames = 200, scene_peak = 0, npos = 0;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--scene-peak"))
            scene_peak = 1;
        else if (npos++ == 0)
            streams = atoi(argv[i]);
        else
            frames = atoi(argv[i]);
    }
    const int devices = pl_hip_device_count();
    if (devices < 1) {
        fprintf(stderr, "bench_streams: no HIP device\n");
        return 1;
    }
    if (streams <= 0)
        streams = devices;
    if (scene_peak && streams > devices) {
        fprintf(stderr, "bench_streams: --scene-peak needs one device per stream (%d devices)\n", devices);
        return 1;
    }

    // HDR10 test frame: the reference's bench pattern (src/tests/bench.c:32-51) as PQ code values,
    // scaled so that the peak is ~1000 cd/m^2 (PQ 0.75), as bench.py's synthetic_frame does
    uint16_t *pixels = malloc((size_t) SW * SH * 4 * sizeof(uint16_t));
    const double xc = (SW - 1) / 2.0, yc = (SH - 1) / 2.0, phi = 1.6180339887498948;
    const double fr = 0.1 * M_PI * 0.5 / sqrt(xc * xc + yc * yc), fg = fr / phi, fb = fg / phi;
    for (int y = 0; y < SH; y++) {
        for (int x = 0; x < SW; x++) {
            const double r2 = (x - xc) * (x - xc) + (y - yc) * (y - yc);
            uint16_t *px = &pixels[4 * ((size_t) y * SW + x)];
            px[0] = (uint16_t) (0.75 * lrint(65535.0 * (0.5 * sin(fr * r2) + 0.5)));
            px[1] = (uint16_t) (0.75 * lrint(65535.0 * (0.5 * sin(fg * r2) + 0.5)));
            px[2] = (uint16_t) (0.75 * lrint(65535.0 * (0.5 * sin(fb * r2) + 0.5)));
            px[3] = 65535;
        }
    }

    // RCCL, bound at run time (the library itself has no link-time dependency on it either)
    void **comms = NULL, *all_reduce = NULL;
    if (scene_peak) {
        void *dl = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!dl)
            dl = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
        int (*init_all)(void **, int, const int *) = dl ? (int (*)(void **, int, const int *)) dlsym(dl, "ncclCommInitAll") : NULL;
        all_reduce = dl ? dlsym(dl, "ncclAllReduce") : NULL;
        comms = calloc(streams, sizeof(void *));
        int *devs = calloc(streams, sizeof(int));
        for (int i = 0; i < streams; i++)
            devs[i] = i;
        if (!init_all || !all_reduce || init_all(comms, streams, devs) != 0) {
            fprintf(stderr, "bench_streams: RCCL not available (ncclCommInitAll)\n");
            return 1;
        }
        free(devs);
    }

    pl_log log = pl_log_create(PL_API_VER, pl_log_params(.log_cb = pl_log_simple, .log_level = PL_LOG_WARN));
    pthread_barrier_t start;
    pthread_barrier_init(&start, NULL, streams);
    struct stream *st = calloc(streams, sizeof(*st));
    pthread_t *threads = calloc(streams, sizeof(*threads));
    for (int i = 0; i < streams; i++) {
        st[i] = (struct stream) {
            .index = i, .device = i % devices, .frames = frames, .log = log, .pixels = pixels,
            .start = &start, .nccl_comm = comms ? comms[i] : NULL, .nccl_all_reduce = all_reduce,
        };
        pthread_create(&threads[i], NULL, stream_main, &st[i]);
    }
    double t0 = 1e300, t1 = 0;
    int errors = 0, exchanges = 0, exchange_errors = 0;
    const char *failure = NULL;
    for (int i = 0; i < streams; i++) {
        pthread_join(threads[i], NULL);
        if (st[i].failure)
            failure = st[i].failure;
        t0 = st[i].t_begin < t0 ? st[i].t_begin : t0;
        t1 = st[i].t_end > t1 ? st[i].t_end : t1;
        errors |= st[i].errors;
        exchanges += st[i].exchanges;
        exchange_errors += st[i].exchange_errors;
    }
    if (failure) {
        fprintf(stderr, "bench_streams: %s\n", failure);
        return 1;
    }
    const double us = t1 - t0;
    printf("{\"program\": \"bench_streams\", \"workload\": \"ewa_1080p_to_4k_hdr_tonemap\", \"streams\": %d, "
           "\"devices\": %d, \"frames_per_stream\": %d, \"ms_per_frame_per_stream\": %.4f, "
           "\"value\": %.1f, \"unit\": \"Mpixels/s\", \"scene_peak_allreduce\": %s, \"peak_exchanges\": %d, "
           "\"exchange_errors\": %d, \"render_errors\": %d}\n",
           streams, devices, frames, us / frames * 1e-3, (double) streams * frames * DW * DH / us,
           scene_peak ? "true" : "false", exchanges, exchange_errors, errors);
    free(pixels);
    pl_log_destroy(&log);
    return errors || exchange_errors ? 2 : 0;
}
