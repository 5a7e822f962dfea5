/*
 * libplacebo-hip — internal helpers shared by the host-side (C) sources.
 */
#ifndef PLH_HOST_COMMON_H_
#define PLH_HOST_COMMON_H_

#include <stdarg.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#include <libplacebo/log.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PL_MIN(a, b) ((a) < (b) ? (a) : (b))
#define PL_MAX(a, b) ((a) > (b) ? (a) : (b))
#define PL_CLAMP(x, lo, hi) (PL_MIN(PL_MAX(x, lo), hi))
#define PL_DEF(x, d) ((x) ? (x) : (d))
#define PL_SQUARE(x) ((x) * (x))
#define PL_ARRAY_SIZE(a) (sizeof(a) / sizeof((a)[0]))
#define PL_ALIGN(x, a) (((x) + (a) - 1) / (a) * (a))
#define PL_ALIGN2(x, a) (((x) + (a) - 1) & ~((a) - 1))

// printf-style logging through an (optional) pl_log
void pl_msg(pl_log log, enum pl_log_level lev, const char *fmt, ...)
    __attribute__((format(printf, 3, 4)));

#ifdef __cplusplus
}
#endif

// the common filter presets, in the order of the reference's COMMON_FILTER_PRESETS (filters.h:28-58):
// shared by pl_filter_presets (filters.c) and pl_scale_filters (renderer.c)
// (a preset's name is its config's: tests/test_host.py)
#define PLH_PRESET(id, desc) {#id, &pl_filter_##id, desc}
#define PLH_COMMON_FILTER_PRESETS \
    PLH_PRESET(bilinear, "Bilinear"), PLH_PRESET(nearest, "Nearest neighbour"), \
    PLH_PRESET(bicubic, "Bicubic"), PLH_PRESET(lanczos, "Lanczos"), \
    PLH_PRESET(ewa_lanczos, "Jinc (EWA Lanczos)"), PLH_PRESET(ewa_lanczossharp, "Sharpened Jinc"), \
    PLH_PRESET(ewa_lanczos4sharpest, "Sharpened Jinc-AR, 4 taps"), PLH_PRESET(gaussian, "Gaussian"), \
    PLH_PRESET(spline16, "Spline (2 taps)"), PLH_PRESET(spline36, "Spline (3 taps)"), \
    PLH_PRESET(spline64, "Spline (4 taps)"), PLH_PRESET(mitchell, "Mitchell-Netravali"), \
    PLH_PRESET(sinc, "Sinc (unwindowed)"), PLH_PRESET(ginseng, "Ginseng (Jinc-Sinc)"), \
    PLH_PRESET(ewa_jinc, "EWA Jinc (unwindowed)"), PLH_PRESET(ewa_ginseng, "EWA Ginseng"), \
    PLH_PRESET(ewa_hann, "EWA Hann"), PLH_PRESET(hermite, "Hermite"), \
    PLH_PRESET(catmull_rom, "Catmull-Rom"), PLH_PRESET(robidoux, "Robidoux"), \
    PLH_PRESET(robidouxsharp, "RobidouxSharp"), PLH_PRESET(ewa_robidoux, "EWA Robidoux"), \
    PLH_PRESET(ewa_robidouxsharp, "EWA RobidouxSharp"),

#endif // PLH_HOST_COMMON_H_
