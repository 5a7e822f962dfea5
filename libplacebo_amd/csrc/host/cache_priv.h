/*
 * libplacebo-hip — cache keys and helpers shared by the users of pl_cache.
 * Key values are the reference's (src/cache.h:61-72), so objects interchange.
 */
#ifndef PLH_CACHE_PRIV_H_
#define PLH_CACHE_PRIV_H_

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#include <libplacebo/cache.h>

enum {
    PLH_CACHE_KEY_SH_LUT    = UINT64_C(0x2206183d320352c6), // host-generated shader LUTs
    PLH_CACHE_KEY_DITHER    = UINT64_C(0x6fed75eb6dce86cb), // blue-noise matrix
    PLH_CACHE_KEY_GAMUT_LUT = UINT64_C(0x6109e47f15d478b1), // gamut-mapping 3D-LUT
};

// SipHash-2-4 under the reference's fixed key (what pl_mem_hash is without xxhash)
uint64_t plh_mem_hash(const void *mem, size_t size);

static inline void plh_hash_merge(uint64_t *accum, uint64_t hash)
{
    *accum ^= hash + UINT64_C(0x9e3779b97f4a7c15) + (*accum << 6) + (*accum >> 2);
}

// Host-generated lookup tables memoised through a pl_cache, with the protocol of the
// reference's sh_lut (src/shaders/lut.c:329,478-486,600): key = CACHE_KEY_SH_LUT ^ signature,
// payload = the table exactly as it is uploaded; an object of another size is a miss.
// Fills `data` (size bytes) from the cache, or through `fill` and then inserts it.
// `cache` may be NULL (always fills). Returns true on a cache hit.
bool plh_cache_memoize(pl_cache cache, uint64_t signature, void *data, size_t size,
                       void (*fill)(void *data, void *priv), void *priv);

#endif // PLH_CACHE_PRIV_H_
