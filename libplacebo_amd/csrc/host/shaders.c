/*
 * libplacebo-hip — pl_shader core: lifetime, signature tracking, persistent
 * objects. Counterpart of the reference's src/shaders.c (pl_shader_alloc :35,
 * pl_shader_reset :91, sh_try_compute :214, sh_bind :513, sh_require :864,
 * object ref-counting :909-963) with GLSL text replaced by typed ops.
 */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "shaders_priv.h"

/* ---- pl_shader_info: the description of a finished shader, shareable beyond its life ---- */

struct info_obj {
    struct pl_shader_info_t pub;
    int refcount;
    char *text;     // the steps, NUL-separated, followed by the joined description
};

static pl_shader_info info_create(pl_shader sh)
{
    struct info_obj *obj = calloc(1, sizeof(*obj));
    if (!obj)
        return NULL;
    const int n = PL_MAX(sh->num_steps, 1);
    size_t len = 0;
    for (int i = 0; i < sh->num_steps; i++)
        len += strlen(sh->steps[i]) + 1;
    // steps + joined description ("a + b + c") + fallback
    obj->text = calloc(1, 2 * len + 3 * n + 32);
    obj->pub.steps = calloc(n, sizeof(char *));
    if (!obj->text || !obj->pub.steps) {
        free(obj->text);
        free((void *) obj->pub.steps);
        free(obj);
        return NULL;
    }
    char *w = obj->text;
    for (int i = 0; i < sh->num_steps; i++) {
        obj->pub.steps[i] = w;
        w = stpcpy(w, sh->steps[i]) + 1;
    }
    obj->pub.num_steps = sh->num_steps;
    obj->pub.description = w;
    if (!sh->num_steps)
        w = stpcpy(w, "(unknown shader)");
    for (int i = 0; i < sh->num_steps; i++) {
        if (i)
            w = stpcpy(w, " + ");
        w = stpcpy(w, sh->steps[i]);
    }
    obj->pub.params = sh->params;
    obj->refcount = 1;
    return &obj->pub;
}

pl_shader_info pl_shader_info_ref(pl_shader_info info)
{
    if (info)
        __atomic_add_fetch(&((struct info_obj *) info)->refcount, 1, __ATOMIC_RELAXED);
    return info;
}

void pl_shader_info_deref(pl_shader_info *pinfo)
{
    struct info_obj *obj = pinfo ? (struct info_obj *) *pinfo : NULL;
    if (!obj)
        return;
    *pinfo = NULL;
    if (__atomic_sub_fetch(&obj->refcount, 1, __ATOMIC_ACQ_REL) > 0)
        return;
    free((void *) obj->pub.steps);
    free(obj->text);
    free(obj);
}

/* ---- finalized shaders that can be turned into a pl_pass ------------------------------------ */
// A small process-wide table: ticket -> live shader. Entries are added by pl_shader_finalize and
// removed when the shader is reset or freed, so a stale "#pl_hip_pass" line resolves to nothing.
#include <pthread.h>

static pthread_mutex_t ticket_lock = PTHREAD_MUTEX_INITIALIZER;
static struct { uint64_t ticket; pl_shader sh; } ticket_tab[128];
static uint64_t ticket_next = 0x1000;

static void ticket_issue(pl_shader sh)
{
    pthread_mutex_lock(&ticket_lock);
    for (size_t i = 0; i < PL_ARRAY_SIZE(ticket_tab); i++) {
        if (!ticket_tab[i].sh) {
            ticket_tab[i].sh = sh;
            ticket_tab[i].ticket = sh->ticket = ++ticket_next;
            break;
        }
    }
    pthread_mutex_unlock(&ticket_lock);
}

static void ticket_revoke(pl_shader sh)
{
    if (!sh->ticket)
        return;
    pthread_mutex_lock(&ticket_lock);
    for (size_t i = 0; i < PL_ARRAY_SIZE(ticket_tab); i++) {
        if (ticket_tab[i].sh == sh)
            ticket_tab[i].sh = NULL;
    }
    pthread_mutex_unlock(&ticket_lock);
    sh->ticket = 0;
}

pl_shader plh_shader_from_glsl(const char *glsl)
{
    const char *tag = glsl ? strstr(glsl, "#pl_hip_pass ") : NULL;
    if (!tag)
        return NULL;
    const uint64_t ticket = strtoull(tag + strlen("#pl_hip_pass "), NULL, 16);
    pl_shader sh = NULL;
    pthread_mutex_lock(&ticket_lock);
    for (size_t i = 0; ticket && i < PL_ARRAY_SIZE(ticket_tab); i++) {
        if (ticket_tab[i].sh && ticket_tab[i].ticket == ticket)
            sh = ticket_tab[i].sh;
    }
    pthread_mutex_unlock(&ticket_lock);
    return sh;
}

static void sh_release(pl_shader sh)
{
    ticket_revoke(sh);
    pl_shader_info_deref(&sh->info);
    for (int i = 0; i < sh->num_held; i++)
        pl_shader_obj_destroy(&sh->held[i]);
    sh->num_held = 0;
    free(sh->errdiff);
    sh->errdiff = NULL;
    if (sh->scratch)
        plh_gpu_release_scratch(sh->scratch_gpu, sh->scratch);
    sh->scratch = NULL;
}

pl_shader pl_shader_alloc(pl_log log, const struct pl_shader_params *params)
{
    struct pl_shader_t *sh = calloc(1, sizeof(*sh));
    if (!sh)
        return NULL;
    sh->log = log;
    pl_shader_reset(sh, params);
    return sh;
}

void pl_shader_free(pl_shader *psh)
{
    pl_shader sh = psh ? *psh : NULL;
    if (!sh)
        return;
    sh_release(sh);
    free(sh->listing);
    free(sh);
    *psh = NULL;
}

void pl_shader_reset(pl_shader sh, const struct pl_shader_params *params)
{
    sh_release(sh);
    pl_log log = sh->log;
    char *listing = sh->listing;
    const size_t cap = sh->listing_cap;
    memset(sh, 0, sizeof(*sh));
    sh->log = log;
    sh->listing = listing;
    sh->listing_cap = cap;
    if (listing)
        listing[0] = '\0';
    sh->mutable_ = true;
    if (params)
        sh->params = *params;
}

bool pl_shader_is_failed(const pl_shader sh) { return sh->failed; }
bool pl_shader_is_compute(const pl_shader sh) { return sh->is_compute; }

bool pl_shader_output_size(const pl_shader sh, int *w, int *h)
{
    if (!sh->output_w || !sh->output_h)
        return false;
    *w = sh->transpose ? sh->output_h : sh->output_w;
    *h = sh->transpose ? sh->output_w : sh->output_h;
    return true;
}

// Every recording entry point starts here: may the shader take another stage that consumes
// `insig` and (optionally) fixes the output size to w x h? On success the shader produces a colour.
bool sh_require(pl_shader sh, enum pl_shader_sig insig, int w, int h)
{
    static const char *const sig_name[] = { "PL_SHADER_SIG_NONE", "PL_SHADER_SIG_COLOR",
                                            "PL_SHADER_SIG_SAMPLER" };
    const char *why = NULL;
    if (sh->failed)
        why = "Attempting to modify a failed shader!";
    else if (!sh->mutable_)
        why = "Attempted to modify an immutable shader!";
    if (why) {
        SH_FAIL(sh, "%s", why);
        return false;
    }

    const bool w_clash = w && sh->output_w && sh->output_w != w;
    const bool h_clash = h && sh->output_h && sh->output_h != h;
    if (w_clash || h_clash) {
        SH_FAIL(sh, "Illegal sequence of shader operations: Incompatible output "
                "size requirements %dx%d and %dx%d", sh->output_w, sh->output_h, w, h);
        return false;
    }

    // an empty shader adopts the stage's input as its own; otherwise the signatures must chain
    if (sh->output == PL_SHADER_SIG_NONE && insig != PL_SHADER_SIG_NONE) {
        sh->input = insig;
    } else if (sh->output != insig) {
        SH_FAIL(sh, "Illegal sequence of shader operations! Current output signature "
                "is '%s', but called operation expects '%s'!", sig_name[sh->output], sig_name[insig]);
        return false;
    }

    sh->output = PL_SHADER_SIG_COLOR;
    if (!sh->output_w)
        sh->output_w = w;
    if (!sh->output_h)
        sh->output_h = h;
    return true;
}

bool sh_try_compute(pl_shader sh, int bw, int bh, bool flex, size_t mem)
{
    (void) flex;
    const struct pl_glsl_version glsl = sh_glsl(sh);
    if (!glsl.compute || sh->shmem + mem > glsl.max_shmem_size)
        return false;
    sh->shmem += mem;
    sh->group_size[0] = bw;
    sh->group_size[1] = bh;
    sh->is_compute = true;
    return true;
}

// Names the stage being recorded: one entry of pl_shader_info.steps (sh_describef in the
// reference appends to the same list, src/shaders.c:140-150)
void sh_describef(pl_shader sh, const char *fmt, ...)
{
    if (sh->num_steps == (int) PL_ARRAY_SIZE(sh->steps))
        return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(sh->steps[sh->num_steps++], sizeof(sh->steps[0]), fmt, ap);
    va_end(ap);
}

const char *sh_description(pl_shader sh)
{
    return sh->num_steps ? sh->steps[sh->num_steps - 1] : "(unknown shader)";
}

void sh_listf(pl_shader sh, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    const int n = vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (n <= 0)
        return;
    const size_t add = PL_MIN((size_t) n, sizeof(buf) - 1);
    if (sh->listing_len + add + 1 > sh->listing_cap) {
        const size_t cap = PL_MAX(sh->listing_cap * 2, sh->listing_len + add + 256);
        char *nl = realloc(sh->listing, cap);
        if (!nl)
            return;
        sh->listing = nl;
        sh->listing_cap = cap;
    }
    memcpy(sh->listing + sh->listing_len, buf, add + 1);
    sh->listing_len += add;
}

struct plh_op *sh_op(pl_shader sh, int kind)
{
    if (sh->pass.num_ops >= PLH_MAX_OPS) {
        SH_FAIL(sh, "Too many colour stages in one shader (max %d)", PLH_MAX_OPS);
        return NULL;
    }
    struct plh_op *op = &sh->pass.ops[sh->pass.num_ops++];
    memset(op, 0, sizeof(*op));
    op->kind = kind;
    return op;
}

const struct pl_shader_res *pl_shader_finalize(pl_shader sh)
{
    if (sh->failed)
        return NULL;
    if (!sh->mutable_)
        return &sh->res;
    if (!sh->info)
        sh->info = info_create(sh);
    if (!sh->info) {
        sh->failed = true;
        return NULL;
    }
    // the op list, "serialised" for pl_pass_create: a ticket that resolves to it for as long as
    // this shader is alive and unchanged (the life time of pl_shader_res itself)
    if (SH_GPU(sh) && sh->kind == PLH_SHADER_PASS) {
        ticket_issue(sh);
        if (sh->ticket)
            sh_listf(sh, "#pl_hip_pass %016llx\n", (unsigned long long) sh->ticket);
    }
    sh->res = (struct pl_shader_res) {
        .info = sh->info,
        .glsl = sh->listing ? sh->listing : "",
        .name = "main",
        .input = sh->input,
        .output = sh->output,
        .compute_group_size = { sh->group_size[0], sh->group_size[1] },
        .compute_shmem = sh->shmem,
        // mirrors of `info` for programs written against older API levels
        .params = sh->info->params,
        .steps = sh->info->steps,
        .num_steps = sh->info->num_steps,
        .description = sh->info->description,
    };
    sh->mutable_ = false;
    return &sh->res;
}

/* ------------------------------------------------------------------------ */
/* persistent objects                                                        */

void pl_shader_obj_destroy(pl_shader_obj *ptr)
{
    pl_shader_obj obj = ptr ? *ptr : NULL;
    if (!obj)
        return;
    *ptr = NULL;
    if (--obj->refcount > 0)
        return;
    if (obj->uninit)
        obj->uninit(obj->gpu, obj->priv);
    free(obj->priv);
    free(obj);
}

void *sh_require_obj(pl_shader sh, pl_shader_obj *ptr, enum pl_shader_obj_type type,
                     size_t priv_size, void (*uninit)(pl_gpu gpu, void *priv))
{
    if (!ptr)
        return NULL;

    pl_shader_obj obj = *ptr;
    if (obj && obj->gpu != SH_GPU(sh)) {
        SH_FAIL(sh, "Passed pl_shader_obj belongs to different GPU!");
        return NULL;
    }
    if (obj && obj->type != type) {
        SH_FAIL(sh, "Passed pl_shader_obj of wrong type! Shader objects must "
                "always be used with the same type of shader.");
        return NULL;
    }
    if (!obj) {
        obj = calloc(1, sizeof(*obj));
        if (!obj)
            return NULL;
        obj->refcount = 1;
        obj->gpu = SH_GPU(sh);
        obj->type = type;
        obj->priv = calloc(1, priv_size);
        obj->uninit = uninit;
        if (!obj->priv) {
            free(obj);
            return NULL;
        }
    }
    *ptr = obj;
    return obj->priv;
}

void sh_hold(pl_shader sh, pl_shader_obj obj)
{
    if (!obj)
        return;
    for (int i = 0; i < sh->num_held; i++) {
        if (sh->held[i] == obj)
            return;
    }
    if (sh->num_held >= (int) PL_ARRAY_SIZE(sh->held)) {
        // the recorded pass points into this object's device memory: without the reference it
        // could be freed under the pass, so the shader is refused rather than run unprotected
        SH_FAIL(sh, "Too many state objects in one shader (%d)", sh->num_held);
        return;
    }
    obj->refcount++;
    sh->held[sh->num_held++] = obj;
}

/* ------------------------------------------------------------------------ */

bool sh_bind(pl_shader sh, pl_tex tex, enum pl_tex_address_mode address_mode,
             const pl_rect2df *rect)
{
    if (pl_tex_params_dimension(tex->params) != 2) {
        SH_FAIL(sh, "Failed binding texture: not a 2D texture!");
        return false;
    }
    if (!tex->params.sampleable) {
        SH_FAIL(sh, "Failed binding texture: texture not sampleable!");
        return false;
    }

    struct plh_sampler_args *s = &sh->pass.s;
    plh_tex_view(tex, &s->src);
    s->address_mode = address_mode;
    sh->src_tex = tex;

    // vertex attribute tex_coord = rect / tex_size at the 4 corners, in the
    // reference's order {x0,y0}, {x1,y0}, {x0,y1}, {x1,y1} (shaders.c:497-502)
    const float sx = 1.0 / tex->params.w, sy = 1.0 / tex->params.h;
    const pl_rect2df full = { .x1 = tex->params.w, .y1 = tex->params.h };
    rect = PL_DEF(rect, &full);
    sh->src_rect = *rect;
    const float x0 = sx * rect->x0, y0 = sy * rect->y0,
                x1 = sx * rect->x1, y1 = sy * rect->y1;
    s->pos[0][0] = x0; s->pos[0][1] = y0;
    s->pos[1][0] = x1; s->pos[1][1] = y0;
    s->pos[2][0] = x0; s->pos[2][1] = y1;
    s->pos[3][0] = x1; s->pos[3][1] = y1;
    s->pt[0] = sx;
    s->pt[1] = sy;
    return true;
}

float plh_fmtf(double v)
{
    char buf[64];
    snprintf(buf, sizeof(buf), "%f", v);
    return strtof(buf, NULL);
}

bool plh_shader_aux_eligible(const pl_shader sh)
{
    if (!sh || sh->failed || sh->kind != PLH_SHADER_PASS || !sh->detect_peak || !sh->src_tex)
        return false;
    const struct plh_pass *p = &sh->pass;
    if (p->s.type != PLH_SAMPLE_NEAREST && p->s.type != PLH_SAMPLE_BILINEAR)
        return false;
    for (int i = 0; i < p->num_ops; i++) {
        const struct plh_op *op = &p->ops[i];
        if (op->kind != PLH_OP_PEAK_DETECT && (op->ptr || op->ptr2))
            return false;   // tables, other planes, feature maps: not tracked across streams
    }
    return true;
}
