/*
 * libplacebo-hip: minimal logging interface. Mirrors the reference's
 * src/include/libplacebo/log.h (pl_log object + level/callback), reduced to
 * what the render hot path needs.
 */
#ifndef LIBPLACEBO_LOG_H_
#define LIBPLACEBO_LOG_H_

#include <libplacebo/config.h>

PL_API_BEGIN

enum pl_log_level {
    PL_LOG_NONE = 0,
    PL_LOG_FATAL,
    PL_LOG_ERR,
    PL_LOG_WARN,
    PL_LOG_INFO,
    PL_LOG_DEBUG,
    PL_LOG_TRACE,
    PL_LOG_ALL = PL_LOG_TRACE,
};

struct pl_log_params {
    void (*log_cb)(void *log_priv, enum pl_log_level level, const char *msg);
    void *log_priv;
    enum pl_log_level log_level;
};

#define pl_log_params(...) (&(struct pl_log_params) { __VA_ARGS__ })

typedef const struct pl_log_t {
    struct pl_log_params params;
} *pl_log;

// Create / destroy a logger. A NULL pl_log is valid everywhere and silent.
PL_API pl_log pl_log_create(int api_ver, const struct pl_log_params *params);
PL_API void pl_log_destroy(pl_log *log);

// Stock callback printing to stderr (log_priv unused)
PL_API void pl_log_simple(void *stream, enum pl_log_level level, const char *msg);

PL_API_END

#endif // LIBPLACEBO_LOG_H_
