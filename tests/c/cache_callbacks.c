/*
 * Test infrastructure: the callback half of the reference's pl_cache scenario
 * (src/tests/cache.c:180-211) -- a `get` callback that serves every miss, a `set` callback that
 * counts insertions and deletions. (ctypes cannot return structs by value from a callback, so
 * this part is C.) Exit status 0 = every check held; prints the first failing line otherwise.
 */
#include <stdio.h>
#include <string.h>

#include <libplacebo/cache.h>

#define CHECK(x) do { if (!(x)) { printf("FAILED line %d: %s\n", __LINE__, #x); return 1; } } while (0)

static pl_cache_obj lookup_parity(void *priv, uint64_t key)
{
    // a deliberately wrong key: the cache must report the one that was asked for
    return (pl_cache_obj) { .key = 0xFFFF, .data = (key & 1) ? "bar" : "foo", .size = 3 };
}

static void count_objects(void *priv, pl_cache_obj obj)
{
    *(int *) priv += obj.size ? 1 : -1;
}

int main(void)
{
    const uint64_t k1 = 0x9c65575f419288f5, k2 = 0x92da969be9b88086, k7 = 0x30c18c962d82e5f5;
    int live = 0;
    pl_cache c = pl_cache_create(pl_cache_params(.get = lookup_parity, .set = count_objects,
                                                 .priv = &live));
    pl_cache_obj a = { .key = k1 }, b = { .key = k2 };
    CHECK(pl_cache_get(c, &a));
    CHECK(a.key == k1 && a.size == 3 && !memcmp(a.data, "bar", 3));
    CHECK(pl_cache_get(c, &b));
    CHECK(b.key == k2 && b.size == 3 && !memcmp(b.data, "foo", 3));
    CHECK(pl_cache_objects(c) == 0 && live == 0);
    CHECK(pl_cache_try_set(c, &a));
    CHECK(pl_cache_try_set(c, &b));
    CHECK(pl_cache_try_set(c, &(pl_cache_obj) { .key = k7, .data = "abcde", .size = 5 }));
    CHECK(pl_cache_objects(c) == 3 && live == 3);
    CHECK(pl_cache_try_set(c, &a));     // emptied by the insertion above: deletes
    CHECK(pl_cache_try_set(c, &b));
    CHECK(pl_cache_objects(c) == 1 && live == 1);
    pl_cache_destroy(&c);
    CHECK(!c);
    puts("ok");
    return 0;
}
