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

#endif // PLH_HOST_COMMON_H_
