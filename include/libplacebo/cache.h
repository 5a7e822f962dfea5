/*
 * libplacebo-hip: pl_cache — a keyed store for generated data that is expensive to
 * recompute. API and on-disk format of the reference's src/include/libplacebo/cache.h
 * (pl_cache_obj :30-45, pl_cache_params :47-81, save/load :118-160, set/get :183-205).
 *
 * On this backend there are no compiled programs to cache; what goes through a pl_cache is
 * the 64x64 blue-noise dither matrix (35 ms to generate) and the gamut-mapping 3D-LUTs
 * (180 ms), under the reference's own keys, so a cache file written by either implementation
 * serves the other. Attach a cache with pl_gpu_set_cache (gpu.h).
 *
 * Thread-safety: Safe. NULL is a valid pl_cache everywhere (= no caching).
 */
#ifndef LIBPLACEBO_CACHE_H_
#define LIBPLACEBO_CACHE_H_

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include <libplacebo/config.h>
#include <libplacebo/common.h>
#include <libplacebo/log.h>

PL_API_BEGIN

typedef struct pl_cache_obj {
    uint64_t key;               // identifies the object
    void *data;                 // size 0 = "no object" (setting one deletes the key)
    size_t size;
    void (*free)(void *data);   // releases `data`; NULL on insertion = the cache copies `data`
} pl_cache_obj;

struct pl_cache_params {
    pl_log log;
    size_t max_object_size;     // 0 = unlimited
    size_t max_total_size;      // 0 = unlimited; oldest objects are dropped first
    // Called after an object was inserted, replaced or deleted through pl_cache_set (not for
    // evictions, not for pl_cache_load). `obj` stays owned by the cache. Must be thread-safe.
    void (*set)(void *priv, pl_cache_obj obj);
    // Called on a miss; the returned object (size 0 = none) becomes the cache's / caller's.
    pl_cache_obj (*get)(void *priv, uint64_t key);
    void *priv;
};

#define pl_cache_params(...) (&(struct pl_cache_params) { __VA_ARGS__ })
PL_API extern const struct pl_cache_params pl_cache_default_params;

typedef const struct pl_cache_t {
    struct pl_cache_params params;
} *pl_cache;

PL_API pl_cache pl_cache_create(const struct pl_cache_params *params);  // never fails
PL_API void pl_cache_destroy(pl_cache *cache);
PL_API void pl_cache_reset(pl_cache cache);         // drop every object (no `set` callbacks)
PL_API int pl_cache_objects(pl_cache cache);
PL_API size_t pl_cache_size(pl_cache cache);        // bytes
PL_API uint64_t pl_cache_signature(pl_cache cache); // order-independent digest of the key set

// Serialise every object through `write` / restore through `read` (false = short read).
// Return the number of objects written / accepted; load returns < 0 for a foreign stream.
PL_API int pl_cache_save_ex(pl_cache cache,
                            void (*write)(void *priv, size_t size, const void *ptr),
                            void *priv);
PL_API int pl_cache_load_ex(pl_cache cache,
                            bool (*read)(void *priv, size_t size, void *ptr),
                            void *priv);

// Memory variants. save returns the size needed (call with size 0 to query).
PL_API size_t pl_cache_save(pl_cache cache, uint8_t *data, size_t size);
PL_API int pl_cache_load(pl_cache cache, const uint8_t *data, size_t size);

static inline void pl_write_file_cb(void *priv, size_t size, const void *ptr)
{
    (void) fwrite(ptr, 1, size, (FILE *) priv);
}

static inline bool pl_read_file_cb(void *priv, size_t size, void *ptr)
{
    return fread(ptr, 1, size, (FILE *) priv) == size;
}

#define pl_cache_save_file(c, file) pl_cache_save_ex(c, pl_write_file_cb, file)
#define pl_cache_load_file(c, file) pl_cache_load_ex(c, pl_read_file_cb,  file)

// Ready-made `set` / `get` callbacks keeping one file per object: `priv` is a path prefix
// (char *), the file name is the prefix + 16 lowercase hex digits of the key. A size-0 object
// removes the file; unreadable or corrupt files are removed and count as misses.
PL_API void pl_cache_set_file(void *path, pl_cache_obj obj);
PL_API pl_cache_obj pl_cache_get_file(void *path, uint64_t key);
#define pl_cache_set_dir pl_cache_set_file
#define pl_cache_get_dir pl_cache_get_file

// Insert (or, with size 0, delete). On success the cache owns the object and `obj->data` /
// `obj->free` are cleared; on failure (object larger than the limits) the caller keeps it.
PL_API bool pl_cache_try_set(pl_cache cache, pl_cache_obj *obj);
PL_API void pl_cache_set(pl_cache cache, pl_cache_obj *obj);    // frees `obj` on failure

// Take the object with `obj->key` out of the cache; the caller then owns it (re-insert it or
// release it with obj->free). On a miss everything but the key is cleared.
PL_API bool pl_cache_get(pl_cache cache, pl_cache_obj *obj);

// Visit every object (do not call into the same cache from `cb`)
PL_API void pl_cache_iterate(pl_cache cache,
                             void (*cb)(void *priv, pl_cache_obj obj),
                             void *priv);

static inline void pl_cache_obj_free(pl_cache_obj *obj)
{
    if (obj->free)
        obj->free(obj->data);
    obj->data = NULL;
    obj->free = NULL;
    obj->size = 0;
}

PL_API_END

#endif // LIBPLACEBO_CACHE_H_
