/*
 * TEST INFRASTRUCTURE ONLY -- the reference's own benchmark (src/tests/bench.c: 1080p rgba16hf ->
 * 1080p rgba16hf, 16 rotating targets, 0.5 s warm-up, 1 s per shader) against the HIP backend.
 * bench.c creates a Vulkan backend in its main(); here the same main() gets pl_hip: the three
 * names it uses are mapped below, nothing else differs (cuts: oracle/_ref/gen/bench_cuts.txt --
 * the film-grain benchmarks).
 */
#include <libplacebo/hip.h>
typedef pl_hip pl_vulkan;
#define pl_vulkan_params(...) NULL
#define pl_vulkan_create(log, params) pl_hip_create(log, params)
#define pl_vulkan_destroy(vk) pl_hip_destroy(vk)
#include "bench_hip.c"      /* oracle/_ref/gen: the reference's bench.c minus the listed cuts */
