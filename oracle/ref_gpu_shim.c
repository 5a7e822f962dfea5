/* TEST INFRASTRUCTURE ONLY -- what the reference's src/gpu.c needs from files that are not built
 * into oracle/_ref/libplref_gpu.so (the pure shader-variable / layout helpers of gpu.c are what the
 * tests call; nothing here is ever reached by them). */
#include <stddef.h>
struct pl_tex_transfer_params;
typedef struct pl_dispatch_t *pl_dispatch;
void pl_dispatch_destroy(pl_dispatch *dp) { (void) dp; }
size_t pl_tex_transfer_size(const struct pl_tex_transfer_params *par) { (void) par; return 0; }
void print_drm_mod(void *log, int lev, unsigned long long mod) { (void) log; (void) lev; (void) mod; }
