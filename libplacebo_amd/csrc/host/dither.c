/*
 * libplacebo-hip — Tier-0 host maths: dither matrices + error diffusion kernels.
 *
 * Fresh implementation of the behaviour of the reference's src/dither.c:
 *   pl_generate_bayer_matrix  (dither.c:36-55)   recursive 2x2 refinement
 *   pl_generate_blue_noise    (dither.c:57-190)  void-and-cluster on a torus with
 *                                                a u64 fixed-point exponential
 *                                                energy kernel, libc rand() ties
 *   error-diffusion kernels   (dither.c:192-317)
 *
 * The matrices are consumed by the dither kernel through an *integer* index
 * path (M[y & (n-1)][x & (n-1)]), so they must be bit-identical to the
 * reference's for bit-exact output (tests/test_tier0_ref.py, with srand(1)).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <libplacebo/dither.h>
#include "host_common.h"

void pl_generate_bayer_matrix(float *data, int size)
{
    // Grow a 1x1 seed: every pass turns each sz×sz block into a 2sz×2sz one
    // whose quadrants are offset by k/(4·sz²) in the order
    //   [0 2]
    //   [3 1]
    data[0] = 0;
    for (int sz = 1; sz < size; sz *= 2) {
        const int quad[4] = { 0, sz * size + sz, sz, sz * size };
        for (int y = 0; y < sz; y++) {
            for (int x = 0; x < sz; x++) {
                const int at = y * size + x;
                for (int k = 1; k < 4; k++)
                    data[at + quad[k]] = data[at] + k / (4.0 * sz * sz);
            }
        }
    }
}

/* ------------------------------------------------------------------------ */
/* void-and-cluster                                                          */

struct vc_state {
    unsigned bits, n, n2;       // n = 1 << bits, n2 = n*n
    uint64_t *kernel;           // n2 energy kernel, window in the top-left
    uint64_t *energy;           // n2 accumulated energy
    uint8_t *placed;            // n2
    uint64_t *rank;             // n2 output ranks
    uint64_t *ties;             // n2 scratch for tie candidates
    unsigned centre;            // flat index of the kernel window centre
};

static void vc_build_kernel(struct vc_state *s)
{
    // (2r+1)² window of exp(-d·sigma), scaled so the sum cannot overflow u64
    const unsigned r = s->n / 2 - 1;
    const unsigned win = 2 * r + 1, win2 = win * win;
    const double sigma = -log(1.5 / (double) UINT64_MAX * win2) / r;

    memset(s->kernel, 0, s->n2 * sizeof(uint64_t));
    for (unsigned gy = 0; gy < win; gy++) {
        for (unsigned gx = 0; gx < win; gx++) {
            const int cx = (int) gx - (int) r, cy = (int) gy - (int) r;
            const int sq = cx * cx + cy * cy;
            const double e = exp(-sqrt(sq) * sigma);
            s->kernel[gx | (gy << s->bits)] = e / win2 * (double) UINT64_MAX;
        }
    }
    s->centre = r | (r << s->bits);
}

// Mark cell c as placed and splat the kernel centred on it. The splat is a
// cyclic shift of the *flattened* array (row carries included), exactly as in
// the reference (dither.c:124-138), not an independent wrap per axis.
static void vc_place(struct vc_state *s, uint64_t c)
{
    if (s->placed[c])
        return;
    s->placed[c] = 1;
    const uint64_t mask = s->n2 - 1;
    const uint64_t shift = (s->centre + s->n2 - c) & mask;
    for (uint64_t i = 0; i < s->n2; i++)
        s->energy[i] += s->kernel[(i + shift) & mask];
}

// Index of the emptiest unplaced cell; ties resolved by rand() over the
// candidates in ascending index order.
static uint64_t vc_largest_void(struct vc_state *s)
{
    uint64_t best = UINT64_MAX;
    unsigned nties = 0;
    for (uint64_t c = 0; c < s->n2; c++) {
        if (s->placed[c])
            continue;
        const uint64_t e = s->energy[c];
        if (e > best)
            continue;
        if (e != best) {
            best = e;
            nties = 0;
        }
        s->ties[nties++] = c;
    }

    if (nties == 1)
        return s->ties[0];
    if (nties == s->n2)
        return s->n2 / 2; // empty board: start in the middle
    return s->ties[rand() % nties];
}

void pl_generate_blue_noise(float *data, int size)
{
    unsigned bits = 0;
    while ((1 << bits) < size)
        bits++;
    if (size <= 1 || (1 << bits) != size || bits > 8) {
        if (size == 1)
            data[0] = 0;
        return;
    }

    struct vc_state s = { .bits = bits, .n = size, .n2 = (unsigned) size * size };
    s.kernel = malloc(s.n2 * sizeof(uint64_t));
    s.energy = calloc(s.n2, sizeof(uint64_t));
    s.placed = calloc(s.n2, 1);
    s.rank   = calloc(s.n2, sizeof(uint64_t));
    s.ties   = malloc(s.n2 * sizeof(uint64_t));

    vc_build_kernel(&s);
    for (uint64_t k = 0; k < s.n2; k++) {
        const uint64_t c = vc_largest_void(&s);
        vc_place(&s, c);
        s.rank[c] = k;
    }

    const float denom = s.n2;
    for (unsigned i = 0; i < s.n2; i++)
        data[i] = s.rank[i] / denom;

    free(s.kernel); free(s.energy); free(s.placed); free(s.rank); free(s.ties);
}

/* ------------------------------------------------------------------------ */
/* error diffusion kernels (weights for dx = -2..2, dy = 0..2)               */

#define EDK(sym, nm, desc, sh, div, r0, r1, r2)                                \
    const struct pl_error_diffusion_kernel sym = {                              \
        .name = nm, .description = desc, .shift = sh, .divisor = div,           \
        .pattern = { r0, r1, r2 } }
#define R(a, b, c, d, e) {a, b, c, d, e}

EDK(pl_error_diffusion_simple, "simple", "Simple error diffusion", 1, 2,
    R(0,0,0,1,0), R(0,0,1,0,0), R(0,0,0,0,0));
EDK(pl_error_diffusion_false_fs, "false-fs", "False Floyd-Steinberg kernel", 1, 8,
    R(0,0,0,3,0), R(0,0,3,2,0), R(0,0,0,0,0));
EDK(pl_error_diffusion_sierra_lite, "sierra-lite", "Sierra Lite kernel", 2, 4,
    R(0,0,0,2,0), R(0,1,1,0,0), R(0,0,0,0,0));
EDK(pl_error_diffusion_floyd_steinberg, "floyd-steinberg", "Floyd Steinberg kernel", 2, 16,
    R(0,0,0,7,0), R(0,3,5,1,0), R(0,0,0,0,0));
EDK(pl_error_diffusion_atkinson, "atkinson", "Atkinson kernel", 2, 8,
    R(0,0,0,1,1), R(0,1,1,1,0), R(0,0,1,0,0));
EDK(pl_error_diffusion_jarvis_judice_ninke, "jarvis-judice-ninke",
    "Jarvis, Judice & Ninke kernel", 3, 48,
    R(0,0,0,7,5), R(3,5,7,5,3), R(1,3,5,3,1));
EDK(pl_error_diffusion_stucki, "stucki", "Stucki kernel", 3, 42,
    R(0,0,0,8,4), R(2,4,8,4,2), R(1,2,4,2,1));
EDK(pl_error_diffusion_burkes, "burkes", "Burkes kernel", 3, 32,
    R(0,0,0,8,4), R(2,4,8,4,2), R(0,0,0,0,0));
EDK(pl_error_diffusion_sierra2, "sierra-2", "Two-row Sierra", 3, 16,
    R(0,0,0,4,3), R(1,2,3,2,1), R(0,0,0,0,0));
EDK(pl_error_diffusion_sierra3, "sierra-3", "Three-row Sierra", 3, 32,
    R(0,0,0,5,3), R(2,4,5,4,2), R(0,2,3,2,0));

const struct pl_error_diffusion_kernel * const pl_error_diffusion_kernels[] = {
    &pl_error_diffusion_simple,
    &pl_error_diffusion_false_fs,
    &pl_error_diffusion_sierra_lite,
    &pl_error_diffusion_floyd_steinberg,
    &pl_error_diffusion_atkinson,
    &pl_error_diffusion_jarvis_judice_ninke,
    &pl_error_diffusion_stucki,
    &pl_error_diffusion_burkes,
    &pl_error_diffusion_sierra2,
    &pl_error_diffusion_sierra3,
    NULL
};

const int pl_num_error_diffusion_kernels = PL_ARRAY_SIZE(pl_error_diffusion_kernels) - 1;

const struct pl_error_diffusion_kernel *pl_find_error_diffusion_kernel(const char *name)
{
    for (int i = 0; name && i < pl_num_error_diffusion_kernels; i++) {
        if (!strcmp(name, pl_error_diffusion_kernels[i]->name))
            return pl_error_diffusion_kernels[i];
    }
    return NULL;
}
