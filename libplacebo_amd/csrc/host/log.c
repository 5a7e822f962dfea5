/*
 * libplacebo-hip — logging (role of the reference's src/log.c: pl_log object
 * with level filter + user callback; NULL log is silent).
 */
#include <stdio.h>
#include <stdlib.h>

#include "host_common.h"

const struct pl_log_params pl_log_default_params = {0};

int pl_fix_ver(void)
{
    return 0;
}

const char *pl_version(void)
{
    return "v7.365.0-hip";
}

// `pl_log_create` is a macro that appends the API level (log.h)
pl_log pl_log_create(int api_ver, const struct pl_log_params *params)
{
    struct pl_log_t *log = calloc(1, sizeof(*log));
    if (!log)
        return NULL;
    if (params)
        log->params = *params;
    if (api_ver != PL_API_VER) {
        pl_msg(log, PL_LOG_WARN, "application built against API v%d, library implements v%d",
               api_ver, PL_API_VER);
    }
    return log;
}

// the plain name as well, for callers that resolve symbols by name (ctypes harness)
#undef pl_log_create
PL_API pl_log pl_log_create(int api_ver, const struct pl_log_params *params);
pl_log pl_log_create(int api_ver, const struct pl_log_params *params)
{
    return pl_log_glue2(pl_log_create_, PL_API_VER)(api_ver, params);
}

struct pl_log_params pl_log_update(pl_log ptr, const struct pl_log_params *params)
{
    struct pl_log_t *log = (struct pl_log_t *) ptr;
    if (!log)
        return (struct pl_log_params) {0};
    const struct pl_log_params old = log->params;
    log->params = params ? *params : (struct pl_log_params) {0};
    return old;
}

enum pl_log_level pl_log_level_update(pl_log ptr, enum pl_log_level level)
{
    struct pl_log_t *log = (struct pl_log_t *) ptr;
    if (!log)
        return PL_LOG_NONE;
    const enum pl_log_level old = log->params.log_level;
    log->params.log_level = level;
    return old;
}

void pl_log_destroy(pl_log *log)
{
    if (log && *log) {
        free((void *) *log);
        *log = NULL;
    }
}

void pl_log_simple(void *stream, enum pl_log_level level, const char *msg)
{
    static const char *tags[] = { "", "fatal", "error", "warn", "info", "debug", "trace" };
    FILE *f = stream ? (FILE *) stream : stderr;
    fprintf(f, "[pl-hip %5s] %s\n", tags[level <= PL_LOG_TRACE ? level : 0], msg);
}

void pl_log_color(void *stream, enum pl_log_level level, const char *msg)
{
    static const char *tags[] = { "", "fatal", "error", "warn", "info", "debug", "trace" };
    static const char *sgr[] = { "0", "1;31", "31", "33", "32", "36", "2" };
    FILE *f = stream ? (FILE *) stream : stderr;
    const int l = level <= PL_LOG_TRACE ? level : 0;
    fprintf(f, "\033[%sm[pl-hip %5s]\033[0m %s\n", sgr[l], tags[l], msg);
}

void pl_msg(pl_log log, enum pl_log_level lev, const char *fmt, ...)
{
    if (!log || !log->params.log_cb || lev > log->params.log_level)
        return;
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    log->params.log_cb(log->params.log_priv, lev, buf);
}
