/*
 * libplacebo-hip — pl_cache (include/libplacebo/cache.h).
 *
 * Behaviour of the reference's src/cache.c (insertion order kept, FIFO eviction under
 * max_total_size, `get` hands the object out of the cache, stream format :283-298); the
 * structure is this implementation's own: an intrusive queue of slots, oldest first.
 *
 * Stream format (little endian, shared with the reference so cache files interchange):
 *   "pl_cache" u32 version=1 u32 count, then per object: u64 key, u64 size, u64 digest,
 *   the bytes, zero padding to a multiple of 4. The digest is SipHash-2-4 under the
 *   reference's fixed key (src/hash.h:110-168; what a libplacebo built without xxhash uses).
 */
#include <limits.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <libplacebo/cache.h>
#include "host_common.h"
#include "cache_priv.h"

/* ---- SipHash-2-4 (Aumasson & Bernstein), 64-bit output ---- */

static inline uint64_t rotl64(uint64_t v, int s)
{
    return (v << s) | (v >> (64 - s));
}

struct sip {
    uint64_t a, b, c, d;
};

static inline void sip_round(struct sip *s)
{
    s->a += s->b; s->b = rotl64(s->b, 13); s->b ^= s->a; s->a = rotl64(s->a, 32);
    s->c += s->d; s->d = rotl64(s->d, 16); s->d ^= s->c;
    s->a += s->d; s->d = rotl64(s->d, 21); s->d ^= s->a;
    s->c += s->b; s->b = rotl64(s->b, 17); s->b ^= s->c; s->c = rotl64(s->c, 32);
}

static inline void sip_absorb(struct sip *s, uint64_t word)
{
    s->d ^= word;
    sip_round(s);
    sip_round(s);
    s->a ^= word;
}

uint64_t plh_mem_hash(const void *mem, size_t size)
{
    if (!size)
        return UINT64_C(0x8533321381b8254b);    // the reference's value for empty input

    const uint64_t k0 = UINT64_C(0xfe9f075098ddb0fa), k1 = UINT64_C(0x68f7f03510e5285c);
    struct sip s = {
        UINT64_C(0x736f6d6570736575) ^ k0, UINT64_C(0x646f72616e646f6d) ^ k1,
        UINT64_C(0x6c7967656e657261) ^ k0, UINT64_C(0x7465646279746573) ^ k1,
    };

    const uint8_t *p = mem;
    size_t left = size;
    for (; left >= 8; left -= 8, p += 8) {
        uint64_t w = 0;
        for (int i = 7; i >= 0; i--)
            w = w << 8 | p[i];
        sip_absorb(&s, w);
    }

    uint64_t last = (uint64_t) size << 56;
    for (size_t i = 0; i < left; i++)
        last |= (uint64_t) p[i] << (8 * i);
    sip_absorb(&s, last);

    s.c ^= 0xff;
    for (int i = 0; i < 4; i++)
        sip_round(&s);
    return s.a ^ s.b ^ s.c ^ s.d;
}

/* ---- the store ---- */

struct slot {
    pl_cache_obj obj;
    struct slot *older, *newer;
};

struct cache {
    struct pl_cache_t pub;
    pthread_mutex_t lock;
    struct slot *oldest, *newest;
    int count;
    size_t bytes;
};

#define CACHE(c) ((struct cache *) (c))

const struct pl_cache_params pl_cache_default_params = {0};

static void *xalloc(size_t size)
{
    void *p = calloc(1, size ? size : 1);
    if (!p)
        abort();
    return p;
}

pl_cache pl_cache_create(const struct pl_cache_params *params)
{
    struct cache *c = xalloc(sizeof(*c));
    pthread_mutex_init(&c->lock, NULL);
    if (params)
        c->pub.params = *params;
    struct pl_cache_params *par = &c->pub.params;
    if (!par->max_total_size)
        par->max_total_size = SIZE_MAX;
    if (!par->max_object_size || par->max_object_size > par->max_total_size)
        par->max_object_size = par->max_total_size;
    return &c->pub;
}

static void unlink_slot(struct cache *c, struct slot *s)
{
    *(s->older ? &s->older->newer : &c->oldest) = s->newer;
    *(s->newer ? &s->newer->older : &c->newest) = s->older;
    c->count--;
    c->bytes -= s->obj.size;
}

static void drop_slot(struct cache *c, struct slot *s)
{
    unlink_slot(c, s);
    if (s->obj.free)
        s->obj.free(s->obj.data);
    free(s);
}

static struct slot *find_slot(struct cache *c, uint64_t key)
{
    for (struct slot *s = c->newest; s; s = s->older) {
        if (s->obj.key == key)
            return s;
    }
    return NULL;
}

void pl_cache_reset(pl_cache cache)
{
    if (!cache)
        return;
    struct cache *c = CACHE(cache);
    pthread_mutex_lock(&c->lock);
    while (c->oldest)
        drop_slot(c, c->oldest);
    pthread_mutex_unlock(&c->lock);
}

void pl_cache_destroy(pl_cache *pcache)
{
    if (!pcache || !*pcache)
        return;
    struct cache *c = CACHE(*pcache);
    pl_cache_reset(*pcache);
    pthread_mutex_destroy(&c->lock);
    free(c);
    *pcache = NULL;
}

int pl_cache_objects(pl_cache cache)
{
    if (!cache)
        return 0;
    struct cache *c = CACHE(cache);
    pthread_mutex_lock(&c->lock);
    const int n = c->count;
    pthread_mutex_unlock(&c->lock);
    return n;
}

size_t pl_cache_size(pl_cache cache)
{
    if (!cache)
        return 0;
    struct cache *c = CACHE(cache);
    pthread_mutex_lock(&c->lock);
    const size_t n = c->bytes;
    pthread_mutex_unlock(&c->lock);
    return n;
}

uint64_t pl_cache_signature(pl_cache cache)
{
    if (!cache)
        return 0;
    struct cache *c = CACHE(cache);
    uint64_t sig = 0;
    pthread_mutex_lock(&c->lock);
    for (struct slot *s = c->oldest; s; s = s->newer)
        sig ^= s->obj.key;
    pthread_mutex_unlock(&c->lock);
    return sig;
}

void pl_cache_iterate(pl_cache cache, void (*cb)(void *priv, pl_cache_obj obj), void *priv)
{
    if (!cache)
        return;
    struct cache *c = CACHE(cache);
    pthread_mutex_lock(&c->lock);
    for (struct slot *s = c->oldest; s; s = s->newer)
        cb(priv, s->obj);
    pthread_mutex_unlock(&c->lock);
}

// lock held. Takes ownership of `obj` iff it returns true.
static bool insert_locked(struct cache *c, pl_cache_obj obj)
{
    struct slot *prev = find_slot(c, obj.key);
    if (prev)
        drop_slot(c, prev);
    if (!obj.size)
        return true;    // a deletion
    const struct pl_cache_params *par = &c->pub.params;
    if (obj.size > par->max_object_size) {
        pl_msg(par->log, PL_LOG_DEBUG, "cache: object %016llx (%zu bytes) over the size limit",
               (unsigned long long) obj.key, obj.size);
        return false;
    }
    while (c->oldest && (c->bytes + obj.size > par->max_total_size || c->count == INT_MAX))
        drop_slot(c, c->oldest);

    struct slot *s = xalloc(sizeof(*s));
    s->obj = obj;
    if (!obj.free) {
        s->obj.data = xalloc(obj.size);
        memcpy(s->obj.data, obj.data, obj.size);
        s->obj.free = free;
    }
    s->older = c->newest;
    *(c->newest ? &c->newest->newer : &c->oldest) = s;
    c->newest = s;
    c->count++;
    c->bytes += obj.size;
    return true;
}

bool pl_cache_try_set(pl_cache cache, pl_cache_obj *pobj)
{
    if (!cache)
        return false;
    struct cache *c = CACHE(cache);
    pl_cache_obj seen = *pobj;      // what the `set` callback gets to look at
    pthread_mutex_lock(&c->lock);
    const bool ok = insert_locked(c, seen);
    pthread_mutex_unlock(&c->lock);
    if (ok) {
        *pobj = (pl_cache_obj) { .key = seen.key };
    } else {
        seen = (pl_cache_obj) { .key = seen.key };  // rejected: reported as a deletion
    }
    if (cache->params.set)
        cache->params.set(cache->params.priv, seen);
    return ok;
}

void pl_cache_set(pl_cache cache, pl_cache_obj *obj)
{
    if (pl_cache_try_set(cache, obj))
        return;
    if (obj->free)
        obj->free(obj->data);
    *obj = (pl_cache_obj) { .key = obj->key };
}

static void keep(void *data)
{
    (void) data;
}

bool pl_cache_get(pl_cache cache, pl_cache_obj *out)
{
    const uint64_t key = out->key;
    *out = (pl_cache_obj) { .key = key };
    if (!cache)
        return false;

    struct cache *c = CACHE(cache);
    pthread_mutex_lock(&c->lock);
    struct slot *s = find_slot(c, key);
    if (s) {
        unlink_slot(c, s);
        *out = s->obj;
        free(s);
    }
    pthread_mutex_unlock(&c->lock);
    if (s)
        return true;

    if (cache->params.get) {
        pl_cache_obj ext = cache->params.get(cache->params.priv, key);
        if (ext.size) {
            ext.key = key;
            if (!ext.free)
                ext.free = keep;
            *out = ext;
            return true;
        }
    }
    return false;
}

/* ---- streams ---- */

static const char magic[8] = { 'p', 'l', '_', 'c', 'a', 'c', 'h', 'e' };
enum { STREAM_VERSION = 1 };

struct stream_head {
    char magic[8];
    uint32_t version;
    uint32_t count;
};

struct stream_entry {
    uint64_t key, size, digest;
};

static inline size_t pad4(size_t n)
{
    return (n + 3) & ~(size_t) 3;
}

static void write_obj(void (*write)(void *, size_t, const void *), void *priv, pl_cache_obj obj)
{
    static const uint8_t zeros[4] = {0};
    const struct stream_entry e = { obj.key, obj.size, plh_mem_hash(obj.data, obj.size) };
    write(priv, sizeof(e), &e);
    write(priv, obj.size, obj.data);
    write(priv, pad4(obj.size) - obj.size, zeros);
}

int pl_cache_save_ex(pl_cache cache, void (*write)(void *priv, size_t size, const void *ptr),
                     void *priv)
{
    if (!cache)
        return 0;
    struct cache *c = CACHE(cache);
    pthread_mutex_lock(&c->lock);
    struct stream_head h = { .version = STREAM_VERSION, .count = c->count };
    memcpy(h.magic, magic, sizeof(magic));
    write(priv, sizeof(h), &h);
    for (struct slot *s = c->oldest; s; s = s->newer)
        write_obj(write, priv, s->obj);
    const int n = c->count;
    pthread_mutex_unlock(&c->lock);
    return n;
}

// one entry + payload; returns a malloc'ed object (size 0 on any failure)
static pl_cache_obj read_obj(pl_log log, bool (*read)(void *, size_t, void *), void *priv)
{
    struct stream_entry e;
    if (!read(priv, sizeof(e), &e))
        goto truncated;
    const uint64_t padded = (e.size + 3) & ~UINT64_C(3);
    if (padded < e.size || padded > SIZE_MAX) {
        pl_msg(log, PL_LOG_WARN, "cache: implausible object size, ignoring the rest");
        return (pl_cache_obj) {0};
    }
    void *buf = malloc(padded ? padded : 1);
    if (!buf)
        return (pl_cache_obj) {0};
    if (!read(priv, padded, buf)) {
        free(buf);
        goto truncated;
    }
    if (plh_mem_hash(buf, e.size) != e.digest) {
        pl_msg(log, PL_LOG_WARN, "cache: checksum mismatch, ignoring the rest");
        free(buf);
        return (pl_cache_obj) {0};
    }
    if (!e.size) {
        free(buf);
        return (pl_cache_obj) { .key = e.key, .free = keep };   // marker: valid but empty
    }
    return (pl_cache_obj) { .key = e.key, .data = buf, .size = e.size, .free = free };

truncated:
    pl_msg(log, PL_LOG_WARN, "cache: stream truncated, ignoring the rest");
    return (pl_cache_obj) {0};
}

int pl_cache_load_ex(pl_cache cache, bool (*read)(void *priv, size_t size, void *ptr), void *priv)
{
    if (!cache)
        return 0;
    struct cache *c = CACHE(cache);
    pl_log log = cache->params.log;
    struct stream_head h;
    if (!read(priv, sizeof(h), &h) || memcmp(h.magic, magic, sizeof(magic))) {
        pl_msg(log, PL_LOG_ERR, "cache: not a cache stream");
        return -1;
    }
    if (h.version != STREAM_VERSION || h.count > INT_MAX)
        return 0;

    int loaded = 0;
    pthread_mutex_lock(&c->lock);
    for (uint32_t i = 0; i < h.count; i++) {
        pl_cache_obj obj = read_obj(log, read, priv);
        if (!obj.free)
            break;              // unreadable: stop here
        if (insert_locked(c, obj)) {
            loaded++;
        } else {
            obj.free(obj.data);
        }
    }
    pthread_mutex_unlock(&c->lock);
    return loaded;
}

struct mem_cursor {
    uint8_t *base;
    size_t pos, cap;
};

static void mem_write(void *priv, size_t size, const void *ptr)
{
    struct mem_cursor *m = priv;
    if (m->pos < m->cap) {
        const size_t n = PL_MIN(size, m->cap - m->pos);
        memcpy(m->base + m->pos, ptr, n);
    }
    m->pos += size;
}

static bool mem_read(void *priv, size_t size, void *ptr)
{
    struct mem_cursor *m = priv;
    if (size > m->cap - m->pos)
        return false;
    memcpy(ptr, m->base + m->pos, size);
    m->pos += size;
    return true;
}

size_t pl_cache_save(pl_cache cache, uint8_t *data, size_t size)
{
    struct mem_cursor m = { data, 0, data ? size : 0 };
    pl_cache_save_ex(cache, mem_write, &m);
    return m.pos;
}

int pl_cache_load(pl_cache cache, const uint8_t *data, size_t size)
{
    struct mem_cursor m = { (uint8_t *) data, 0, size };
    return pl_cache_load_ex(cache, mem_read, &m);
}

/* ---- one file per object ---- */

static void object_path(char *out, size_t len, const char *prefix, uint64_t key)
{
    snprintf(out, len, "%s%016llx", prefix, (unsigned long long) key);
}

// File layout = the reference's (src/cache.c:474-521): header, ONE entry, exactly `size` payload
// bytes -- no padding, unlike an entry inside a stream -- so that a directory can be shared with
// other builds of libplacebo.
void pl_cache_set_file(void *priv, pl_cache_obj obj)
{
    const char *dir = priv;
    if (!dir || !dir[0])
        return;
    char name[4096], tmp[4096 + 32];
    object_path(name, sizeof(name), dir, obj.key);
    if (!obj.size) {
        unlink(name);
        return;
    }
    // an existing file is left alone (it is validated when it is read)
    if (access(name, F_OK) == 0)
        return;
    // written under a private name and renamed: a concurrent reader sees the whole file or none
    snprintf(tmp, sizeof(tmp), "%s.%ld.tmp", name, (long) getpid());
    FILE *f = fopen(tmp, "wb");
    if (!f)
        return;
    struct stream_head h = { .version = STREAM_VERSION, .count = 1 };
    memcpy(h.magic, magic, sizeof(magic));
    const struct stream_entry e = { obj.key, obj.size, plh_mem_hash(obj.data, obj.size) };
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(&e, sizeof(e), 1, f) == 1 &&
              fwrite(obj.data, 1, obj.size, f) == obj.size;
    ok = fclose(f) == 0 && ok;
    if (!ok || rename(tmp, name) != 0)
        unlink(tmp);
}

pl_cache_obj pl_cache_get_file(void *priv, uint64_t key)
{
    const char *dir = priv;
    if (!dir || !dir[0])
        return (pl_cache_obj) {0};
    char name[4096];
    object_path(name, sizeof(name), dir, key);
    FILE *f = fopen(name, "rb");
    if (!f)
        return (pl_cache_obj) {0};

    struct stream_head h;
    struct stream_entry e;
    void *data = NULL;
    bool ok = fread(&h, sizeof(h), 1, f) == 1 && !memcmp(h.magic, magic, sizeof(magic)) &&
              h.version == STREAM_VERSION && h.count == 1 &&
              fread(&e, sizeof(e), 1, f) == 1 && e.key == key && e.size && e.size <= SIZE_MAX;
    if (ok) {
        // exactly e.size bytes; whatever follows them (padding written by another version) is ignored
        data = malloc(e.size);
        ok = data && fread(data, 1, e.size, f) == e.size && plh_mem_hash(data, e.size) == e.digest;
    }
    fclose(f);
    if (!ok) {
        free(data);
        unlink(name);   // stale or corrupt
        return (pl_cache_obj) {0};
    }
    return (pl_cache_obj) { .key = key, .data = data, .size = e.size, .free = free };
}

bool plh_cache_memoize(pl_cache cache, uint64_t signature, void *data, size_t size,
                       void (*fill)(void *data, void *priv), void *priv)
{
    pl_cache_obj obj = { .key = PLH_CACHE_KEY_SH_LUT ^ signature };
    if (cache && signature && pl_cache_get(cache, &obj) && obj.size == size) {
        memcpy(data, obj.data, size);
        pl_cache_set(cache, &obj);      // pl_cache_get took it out: hand it back
        return true;
    }
    pl_cache_obj_free(&obj);
    fill(data, priv);
    if (cache && signature) {
        obj = (pl_cache_obj) { .key = PLH_CACHE_KEY_SH_LUT ^ signature, .data = data, .size = size };
        pl_cache_set(cache, &obj);      // free == NULL: the cache keeps its own copy
    }
    return false;
}
