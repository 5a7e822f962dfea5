/*
 * libplacebo-hip: minimal logging interface. Mirrors the reference's
 * src/include/libplacebo/log.h (pl_log object + level/callback), reduced to
 * what the render hot path needs.
 */
#ifndef LIBPLACEBO_LOG_H_
#define LIBPLACEBO_LOG_H_

#include <libplacebo/config.h>
#include <libplacebo/common.h>

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
PL_API extern const struct pl_log_params pl_log_default_params;

typedef const struct pl_log_t {
    struct pl_log_params params;
} *pl_log;

// Create / destroy a logger. A NULL pl_log is valid everywhere and silent. The symbol carries
// the API level, so that a program built against another API level fails to link instead of
// passing mismatched structs (same scheme as the reference, log.h:74-84).
#define pl_log_glue1(x, y) x##y
#define pl_log_glue2(x, y) pl_log_glue1(x, y)
#define pl_log_create pl_log_glue2(pl_log_create_, PL_API_VER)
PL_API pl_log pl_log_create(int api_ver, const struct pl_log_params *params);
PL_API void pl_log_destroy(pl_log *log);

// Swap the callback / level of a live logger; both return the previous setting
PL_API struct pl_log_params pl_log_update(pl_log log, const struct pl_log_params *params);
PL_API enum pl_log_level pl_log_level_update(pl_log log, enum pl_log_level level);

// Stock callbacks printing to a FILE * passed as log_priv (NULL = stderr); the second one
// colours the level tag with ANSI escapes
PL_API void pl_log_simple(void *stream, enum pl_log_level level, const char *msg);
PL_API void pl_log_color(void *stream, enum pl_log_level level, const char *msg);

PL_API_END

#endif // LIBPLACEBO_LOG_H_
