/*
 * libplacebo-hip — logging (role of the reference's src/log.c: pl_log object
 * with level filter + user callback; NULL log is silent).
 */
#include <stdio.h>
#include <stdlib.h>

#include "host_common.h"

pl_log pl_log_create(int api_ver, const struct pl_log_params *params)
{
    (void) api_ver;
    struct pl_log_t *log = calloc(1, sizeof(*log));
    if (log && params)
        log->params = *params;
    return log;
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
