/*
 * libplacebo-hip — internal ABI between the C host layer (shader recording,
 * dispatch, renderer) and the HIP kernels. Plain C structs, passed to the
 * kernels *by value* as kernel arguments (SGPR-resident, wave-uniform).
 *
 * One `plh_pass` describes one GPU pass = what the reference builds as a
 * pl_shader and hands to pl_dispatch_finish (src/dispatch.c:1199):
 *     sampler (how `color` is produced from the source texture)
 *  -> a chain of per-pixel colour ops (what pl_shader_* calls append)
 *  -> store to the target rect (translate_compute_shader, dispatch.c:1079-1144)
 */
#ifndef PLH_DEVICE_H_
#define PLH_DEVICE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- texture formats (packed, linear, pitched 2D arrays in HBM) ---------- */
enum plh_fmt {
    PLH_FMT_NONE = 0,
    PLH_FMT_R8, PLH_FMT_RG8, PLH_FMT_RGBA8,         // unorm8
    PLH_FMT_R16, PLH_FMT_RG16, PLH_FMT_RGBA16,      // unorm16
    PLH_FMT_R16F, PLH_FMT_RG16F, PLH_FMT_RGBA16F,   // IEEE half
    PLH_FMT_R32F, PLH_FMT_RG32F, PLH_FMT_RGBA32F,   // float
    PLH_FMT_COUNT
};

struct plh_view {
    void *ptr;          // device pointer to texel (0,0)
    int32_t w, h;
    int32_t pitch;      // bytes per row
    int32_t fmt;        // enum plh_fmt
};

/* ---- samplers ------------------------------------------------------------ */
enum plh_sampler {
    PLH_SAMPLE_NONE = 0,    // colour starts as (0,0,0,1) (no source)
    PLH_SAMPLE_NEAREST,     // sampling.c:290 (and :277 on non-LINEAR formats)
    PLH_SAMPLE_BILINEAR,    // sampling.c:304 / :277 (hardware bilinear, done in ALU)
    PLH_SAMPLE_BICUBIC,     // sampling.c:318
    PLH_SAMPLE_HERMITE,     // sampling.c:366
    PLH_SAMPLE_GAUSSIAN,    // sampling.c:392
    PLH_SAMPLE_OVERSAMPLE,  // sampling.c:436
    PLH_SAMPLE_POLAR,       // sampling.c:587 (EWA, LDS-tiled)
    PLH_SAMPLE_ORTHO,       // sampling.c:950 (one separable pass)
    PLH_SAMPLE_DEBAND,      // sampling.c:183
    PLH_SAMPLE_DEINTERLACE, // shaders/deinterlacing.c:26 (k_deinterlace.hip)
    PLH_SAMPLE_DISTORT,     // sampling.c:1108 (pl_shader_distort: an affine map of the canvas)
};

enum plh_address_mode {     // gpu.h pl_tex_address_mode
    PLH_ADDRESS_CLAMP = 0,
    PLH_ADDRESS_REPEAT,
    PLH_ADDRESS_MIRROR,
};

/* polar tap: signed offsets relative to the base texel + flags
 * (polar_sample(), sampling.c:503-558) packed as x | y<<8 | flags<<16 */
#define PLH_TAP_SKIPPABLE 1u    // needs the run-time `d < radius` test
#define PLH_TAP_AR        2u    // may contribute to anti-ringing
#define PLH_TAP_PACK(x, y, fl) \
    ((uint32_t) ((uint8_t) (int8_t) (x)) | ((uint32_t) ((uint8_t) (int8_t) (y)) << 8) | \
     ((uint32_t) (fl) << 16))

// POLAR phase-class tables (k_polar.hip, k_polar_pp): the weights of a polar tap
// depend on the output pixel only through fcoord = fract(pos*size - 0.5). For an
// axis-aligned rect fcoord.x is a function of the output column and fcoord.y of
// the row, and over a whole frame only a few dozen distinct fp32 values occur
// (2 canonical phases x fp32 rounding noise for a 2x upscale). The host builds,
// once per geometry, the exact per-(class pair, tap) weights with the same device
// arithmetic the per-pixel path uses; the kernel then only does the FMAs.
#define PLH_PP_LMAX 32      // distinct classes per tile column / row
struct plh_polar_pp {
    int32_t n;              // output pixels per lane and axis (share one base texel)
    int32_t padx, pady;     // cell c covers outputs [n*c - pad, n*c - pad + n)
    int32_t cells_w, cells_h;
    int32_t ncx, ncy;       // number of global classes per axis
    int32_t ntaps;          // compacted tap count
    int32_t tp;             // floats per class pair: ntaps weights + norm (+ padding)
    const float *colfc, *rowfc;         // [width], [height]: fcoord of every column / row
    const int32_t *colbase, *rowbase;   // base texel of every column / row
    const uint8_t *colloc, *rowloc;     // class index local to the tile column / row
    const uint16_t *collist, *rowlist;  // [tiles][PLH_PP_LMAX] local -> global class
    const int32_t *coln, *rown;         // [tiles] number of local classes (dwords: scalar loads)
    const int32_t *colorg, *roworg;     // [tiles] LDS tile origin (texels)
    const float *weights;               // [ncy][ncx][tp]
    const int32_t *tapoff;              // [ntaps] byte offset of the tap in the LDS tile
    const uint32_t *tilemap;            // [launch id] -> tile x | tile y << 16 (XCD-aware
                                        // order), NULL = launch order
};

// POLAR on the matrix pipe (k_polar_mx.hiph): an exact 2x upscale has two phases per axis, so a
// 16-row x 16-column block of same-phase outputs is a small GEMM  out[m][n] = sum_k A[m][k] B[k][n]
// with A = rows of the f16 source tile (LDS, channel-planar) and B = a banded (Toeplitz) matrix
// of the filter weights, split into f16 hi + lo halves (the products are exact in fp32, the sums
// are fp32), plus the first-order terms in the per-pixel phase deviation: B' = d B / d fcoord_x
// and d B / d fcoord_y, scaled by 2^-PLH_MX_DSHIFT, against dfx[] / dfy[] scaled by 2^+PLH_MX_DSHIFT.
// One v_mfma_f32_16x16x32_f16 covers two source rows x 16 source columns.
// The host builds B once per (filter, geometry) in fragment order: frag f, lane l, element e,
// f = 4 * (py * npairs + j) + {hi, lo, d/dx, d/dy} for row pair j of row phase py. A row phase
// only contracts the tap rows that carry a weight at all: the footprint of the reference's tap
// list is 8 rows (taps -3 .. 4, sampling.c:510-515), but at the phases of a centred 2x upscale
// (fcoord 1/4 and 3/4) rows -3 and 4 lie 3.25 and 3.75 texels away, beyond every radius <= 3.25
// (ewa_lanczos: 3.2383) -- six rows = THREE row pairs per phase, each phase starting at its own
// tile row (row_first[py], for the output row pair m = 0), where the shared pairing of round 3
// (py = 0: pairs 0..3, py = 1: pairs 0..4, nine in all) spent a third of its MFMAs on rows of
// zeros. npairs = 3, or 4 for phases / radii with seven or eight live rows.
// An exact 2 : 1 DOWNSCALE (enabled == 2, k_polar_mxd.hip) has ONE phase per axis, fcoord = 1/2, and
// a 14 x 14 footprint that moves two texels per output: out[m][n] = sum_j sum_k S[2 m + j][k] *
// T_j[k - 2 n], K = 44 source columns = two 32-column blocks per source row j. The weights at
// fcoord = (1/2, 1/2) are symmetric in j (T_j = T_13-j, d/dx too, d/dy antisymmetric), so only rows
// j < 7 are stored: frag f = 4 * (2 * j + kb) + {hi, lo, d/dx, d/dy}.
#define PLH_MXD_NFRAG 56
#define PLH_MXD_TAPS 14     // taps per axis, offsets -6 .. 7
#define PLH_MX_NFRAG 32     // 2 row phases x (at most) 4 row pairs x {hi, lo, d/dx, d/dy}
#define PLH_MX_DSHIFT 11
#define PLH_MX_PAD 128      // dfx / dfy are padded to a multiple of this many outputs (the widest tile)
// An exact INTEGER upscale by R = 3 or 4 (enabled == 3, k_polar_mxr.hip) has R phases per axis;
// output column X belongs to base index (X + sx) / R and phase (X + sx) % R (rows: sy), every phase
// of a base shares the base texel, so each row phase py is a GEMM over the same four row pairs.
// A wave owns 8 base columns = two 16-column halves h of 4 bases x R phases (4 R <= 16 columns
// used); the 128 fragments of R = 4 do not fit the LDS next to the tile, so they are staged per
// row phase: frag f = 4 * (2 * (4 * py + j) + h) + {hi, lo, d/dx, d/dy}, 32 per row phase.
// The 3 : 2 upscale (720p -> 1080p, 1440p -> 4K) is the same with a base index standing for a GROUP
// of two source texels (ratio 3, group 2): the outputs of a group start at texel offsets 0 .. 2 of
// it (in the weights' placement only), a base's footprint is 10 rows = five row pairs, a wave's 8
// source columns are 4 groups = ONE half: frag f = 4 * (5 * py + j) + kind, 20 per row phase.
#define PLH_MXR_FRAGS_PER_PHASE 32      // (the stride of a row phase in the blob, either way)
#define PLH_MXR_MAX_RATIO 4
struct plh_polar_mx {
    int32_t enabled;        // 1: the 2x upscale (k_polar_mx), 2: the 2 : 1 downscale (k_polar_mxd),
                            // 3: an upscale by ratio : group, 3, 4 or 3 : 2 (k_polar_mxr)
    int32_t ratio, sx, sy;  // enabled == 3
    int32_t group, pad_;    // enabled == 3: source texels per base index (1, or 2 for 3 : 2)
    int32_t org_x, org_y;   // source texel held by LDS tile (0, 0) of workgroup tile (0, 0)
    int32_t npairs;         // enabled == 1: row pairs per row phase (3 or 4)
    int32_t row_first[2];   // enabled == 1: tile row of the first pair's first row, output row pair 0
    int32_t pad2_;
    void *sink;             // enabled == 1: 512 bytes nobody reads (k_polar_mxp.hip: where the lanes
                            // outside the target put their store)
    const void *bfrag;      // device: [<= PLH_MX_NFRAG][64 lanes][8] f16
    const float *dfx, *dfy; // device: phase deviation of every output column / row, x 2^PLH_MX_DSHIFT
};

struct plh_sampler_args {
    int32_t type;           // enum plh_sampler
    struct plh_view src;
    // vertex attribute `tex_coord` at the 4 quad corners (sh_bind, shaders.c:541-561),
    // in normalised texture coordinates
    float pos[4][2];
    float pt[2];            // 1/tex_size
    int32_t address_mode;
    float scale;            // multiplied into the sampled colour
    uint32_t comp_mask;     // components actually sampled
    int32_t linear;         // texture bound with LINEAR filtering (for deband etc.)
    // BILINEAR only: |rect| in texels, and whether the rect starts on the texel grid. The
    // dispatch lowers a 1:1 on-grid fetch to NEAREST once the output size is known.
    float rect_w, rect_h;
    int32_t rect_on_grid;

    // POLAR: 256-entry radial LUT, as {L[i], L[i+1]} pairs; tap list
    const float *lut;       // device, 2*256 floats (pairs)
    const uint32_t *taps;   // device, packed taps in evaluation order
    int32_t num_taps;
    int32_t bound;          // ceil(radius)
    float radius, rcp_radius, radius_zero;
    float antiring;
    int32_t tile_w, tile_h; // LDS tile (texels)
    int32_t tile_rows;      // output rows per lane (output tile = 32 x 8*rows)
    int32_t tile_fp32;      // tile kept as float4 instead of half4
    const struct plh_polar_pp *pp;  // device; NULL = per-pixel weights
    struct plh_polar_pp ppv;        // the same by value: kernel arguments, one memory round
                                    // trip less at the head of every workgroup
    int32_t pp_lds_weights; // bytes of LDS for the staged weight sub-table
    int32_t pp_n, pp_cells_w, pp_cells_h;   // host copies of pp->n, cells_w, cells_h
    int32_t pp_debug;       // profiling aid (PL_HIP_PP_DEBUG): 1 = no taps, 2 = no verify, 4 = no store
    struct plh_polar_mx mx; // matrix-pipe variant (enabled = 0: k_polar_pp)

    // ORTHO: weights[256][row_stride] rows; N taps along `dir`
    const float *weights;   // device
    int32_t row_size, row_stride;
    int32_t dir;            // 0 = horizontal, 1 = vertical
    int32_t use_linear;     // "linear trick" packing (sampling.c:919-936)
    int32_t use_ar;

    // OVERSAMPLE
    float ratio[2], threshold;

    // DEBAND
    int32_t iterations;
    float db_threshold, db_radius, db_grain;
    float db_neutral[3];
    uint32_t prng_seed;     // frame index (sh_prng, shaders.c:965-998)
    int32_t db_lds;         // the backend's shared-memory limit admits k_deband_lds' window (52 KiB)
};

/* ---- per-pixel colour ops -------------------------------------------------- */
enum plh_op_kind {
    PLH_OP_NONE = 0,
    PLH_OP_SCALE,           // color *= f[0..3]
    PLH_OP_AFFINE,          // color.rgb = M(f[0..8], row-major) * color.rgb + f[9..11]
    PLH_OP_LINEARIZE,       // i0 = transfer; f[] = parameters   (colorspace.c:589)
    PLH_OP_DELINEARIZE,     // i0 = transfer; f[] = parameters   (colorspace.c:722)
    PLH_OP_SIGMOIDIZE,      // f[0]=center f[1]=slope f[2]=offset f[3]=scale (colorspace.c:851)
    PLH_OP_UNSIGMOIDIZE,    // (colorspace.c:874)
    PLH_OP_PREMULTIPLY,     // color.rgb *= color.a  (pl_shader_set_alpha, colorspace.c:26)
    PLH_OP_UNPREMULTIPLY,   // color.rgb /= max(color.a, 1e-6)
    PLH_OP_ALPHA_ONE,       // color.a = 1
    PLH_OP_QUANT_F16,       // round through IEEE half (an rgba16hf FBO store+load)
    PLH_OP_QUANT_UNORM,     // round through unorm of i0 bits (unorm FBO)
    PLH_OP_DITHER,          // ptr = size×size float matrix; i0 = size; i1 = method;
                            // f[0] = 2^depth-1; f[1] = gamma; i2 = depth; f[4..7] = temporal mat2;
                            // ptr2 = the matrix transposed, or NULL (a white-noise plane has none)
    PLH_OP_SWIZZLE,         // i0..i3 packed: output component c takes input comp map[c] (or -1 → 0/1)
    PLH_OP_CLAMP01,         // color = clamp(color, 0, 1)
    PLH_OP_BT2020C_DEC,     // BT.2020 constant luminance (colorspace.c:312-342)
    PLH_OP_BT2020C_ENC,     //                            (colorspace.c:475-493)
    PLH_OP_ICTCP_DEC,       // i0: 0 = PQ, 1 = HLG; f[] = curve constants (colorspace.c:344-390)
    PLH_OP_ICTCP_ENC,       //                                            (colorspace.c:495-524)
    PLH_OP_GAMMA,           // color.rgb = f[0] ? pow(max(color.rgb, 0), f[0]) : 0 (colorspace.c:447-456)
    // colour mapping (colorspace.c:1612-2024), four ops sharing `aux` = i_orig:
    PLH_OP_RGB2IPT,         // f[0..8] = rgb2lms, f[9..13] = 203/10000, m1, c1, c2, c3; f[14] = m2
    PLH_OP_TONE_MAP,        // i0 = mode (0 clip, 1 linear, 2 LUT); f[] see k; ptr = LUT; i1 = size
                            // contrast recovery (:1880-1921): ptr2 = r16hf feature map,
                            // i2 = w | h << 16 (0 = off), f[4] = pitch (int bits),
                            // f[5] = strength, f[6], f[7] = output min / max;
                            // mode 2: f[8] = size - 1, f[9] = size - 2 (as floats)
    PLH_OP_GAMUT_LUT,       // ptr = rgba16 3-D LUT; i0,i1,i2 = sizes; f[0]=scale f[1]=offset f[2]=0.5/pi
                            // f[3] = tricubic; f[4..6] = size - 1, f[7..9] = size - 2, f[10], f[11] = size_I, size_C (as floats)
    PLH_OP_IPT2RGB,         // f[0..8] = lms2rgb, f[9..14] = 1/m2, c1, c2, c3, 1/m1, 10000/203
    PLH_OP_PEAK_DETECT,     // see k_peak.hip; only valid in the 16x16 peak kernel
    // renderer glue (renderer.c)
    PLH_OP_PLANE_MAP,       // color = f[0..3]; color[map[c]] = tmp[c], c < i1; i0 packs map (0xff = none)
    PLH_OP_BLEND_BG,        // color += (1 - color.a) * f[0..3]   (renderer.c:2722-2728)
    // Another plane of the same frame, sampled at this output position (the reference's
    // sh_subpass of a plane shader into pass_read_image, renderer.c:1874-1891):
    //   color[map[c]] = f[8] * texel[c], c < comps. ptr = texels; i0 = w | h << 16; i1 = pitch;
    //   i2 = fmt | comps << 8 | linear << 12 | address << 13 | on_grid << 15 | map << 16 (4 x 4
    //   bits, 0xf = none); f[0..7] = tex_coord at the 4 corners; f[9], f[10] = |rect| in texels
    PLH_OP_PLANE_FETCH,
    // frame mixing (pl_render_image_mix, renderer.c:3944-3993): a second colour register
    PLH_OP_MIX_ADD,         // mix_color += f[0] * color   (mix_color starts at 0)
    PLH_OP_MIX_END,         // color = mix_color
    // pl_shader_extract_features (colorspace.c:1383-1404): color = (I of IPT, 0, 0, 1);
    // f[0..8] = (203/10000) * rgb2lms, f[9..13] = m1 c1 c2 c3 m2
    PLH_OP_FEATURES,
    // pl_shader_custom_lut (shaders/lut.c:212-280): ptr = rgba32f texels, i0 i1 i2 = sizes
    // (1D: i1 = i2 = 0 -> per-channel linear lookup; 3D -> tetrahedral interpolation)
    PLH_OP_CUSTOM_LUT,
    // blend against the tile pattern (renderer.c:2734-2756): outcoord = gl_FragCoord.xy * f[8];
    // tile = lessThan(fract(outcoord), 0.5); color.rgb += (1 - color.a) * (tile.x == tile.y ?
    // f[0..2] : f[4..6]); color.a = 1
    PLH_OP_BLEND_TILES,
    // Dolby Vision (colorspace.c:51-271, :392-420), only in the generic pass kernel's DOVI variant:
    // ptr = struct plh_dovi_comp[3] (device): per component a piecewise polynomial / MMR curve
    PLH_OP_DOVI_RESHAPE,
    // PQ EOTF, f[0..8] = LMS -> RGB (row-major), PQ OETF; f[9..13] = 1/m2 c1 c2 c3 1/m1,
    // f[14], f[15] = m1, m2
    PLH_OP_DOVI_LMS,
};

// pl_reshape_data as the reshaping stage reads it (pl_shader_dovi_reshape packs the same)
struct plh_dovi_comp {
    int32_t num_pivots;         // 0: the component passes through
    int32_t has_poly, has_mmr, mmr_single;
    int32_t min_order, max_order;
    float lo, hi;               // the outer pivots: the result is clamped to them
    float pivots[8];            // the inner pivots, then 1e9 (7 used)
    float coeffs[8][4];         // per piece: polynomial x^0 x^1 x^2, 0 | MMR constant, first row, -, order
    float mmr[48][4];           // per MMR piece and order: two rows (xyz-, then the cross terms)
};

// flags in plh_op.i1 of LINEARIZE / DELINEARIZE / PEAK_DETECT
#define PLH_TRC_CLAMP0   1  // color.rgb = max(color.rgb, 0)
#define PLH_TRC_RESCALE  2  // linearize: scale_out; delinearize: black-scaling prologue

#define PLH_OP_NF 16
struct plh_op {
    int32_t kind;
    int32_t i0, i1, i2;
    float f[PLH_OP_NF];
    const void *ptr;
    const void *ptr2;
};

#define PLH_MAX_OPS 20
#define PLH_PEAK_WORDS 816
#define PLH_PEAK_COPIES 64

// Fused epilogue of the polar kernel for the renderer's common output tail
// ([dither (LUT, power-of-two matrix, not temporal)] [uniform scale] -> rgba16 store):
// parameters at fixed offsets, so they live in SGPRs for the whole kernel instead of being
// decoded by the op interpreter for every pixel. Filled by plh_launch_polar.
struct plh_fast_epi {
    int32_t enabled;
    int32_t has_dither, has_scale;
    int32_t size, mask;         // dither matrix size (power of two) and size - 1
    const float *matrix;
    const float *matrix_t;      // the matrix transposed (column-owning kernels: k_polar_mx), or NULL
    float dscale, dinv;         // 2^depth - 1 and its reciprocal
    float scale;                // color *= scale
    int32_t has_alpha;          // color.a = alpha before the dither (identity PLANE_MAP of rgb)
    float alpha;
};

// The recorded post-ops of the renderer's final passes have one shape (the colour management
// between the scaler and the encoder, renderer.c:2935-3040):
//   [PLANE_MAP] [UNSIGMOIDIZE] [LINEARIZE] [RGB2IPT [TONE_MAP] [GAMUT_LUT] IPT2RGB] [DELINEARIZE]
//   [SIGMOIDIZE] + the fused epilogue above (rgba16 target) or nothing (rgba16hf intermediate)
// -- the HDR map pass (LINEARIZE .. DELINEARIZE), the SDR presets' last scaler pass (UNSIGMOIDIZE
// DELINEARIZE) and their first pass (PLANE_MAP LINEARIZE SIGMOIDIZE into the intermediate). Kernels with a CHAIN variant run exactly these device functions as straight-line
// code -- op indices here, -1 = absent -- instead of walking the interpreter, whose register budget
// is that of its largest op: the same arithmetic in the same order (bit-identical), 14 % fewer
// instructions and half the registers. Filled by the launchers (fastepi.hiph).
struct plh_map_chain {
    int32_t enabled;
    int32_t lin, in, tone, gamut, out, delin;
    int32_t contrast_recovery;  // the tone op reads a feature map (its i2 != 0)
    int32_t unsig, sig;
    int32_t pmap;               // a leading identity PLANE_MAP (missing components := neutral)
    int32_t tail;               // first op of the fused epilogue
    // The uniform scale factors between the chain's stages, folded into its two matrices by the
    // launcher (fastepi.hiph) instead of being multiplied into every pixel (valid when in >= 0):
    //   in_mat  = rgb2ipt.f[9] (203 / 10000) * rgb2lms, times the PQ linearisation's own 10000 / 203
    //             when that is the stage in front (pq_front: the device then leaves it out);
    //   out_mat = ipt2rgb.f[14] (10000 / 203) * lms2rgb, times the delinearisation's black-scaling
    //             slope f[0], with its offset f[1] in out_add (out_rescaled: the device then skips
    //             that step of op_delinearize).
    int32_t pq_front, out_rescaled;
    float in_mat[9], out_mat[9], out_add;
    // The PQ pair of the colour map (and of a pq_front linearisation) as piecewise cubics in LDS
    // (pqseg.hiph). The matcher sets pq_seg = 1 when the chain has the stages and their constants
    // agree, and leaves the constants { m1, c3, m2, 1 / m2, 1 / m1 }; a launcher whose kernel has the
    // variant asks for the device's tables (plh_pqseg_tables) and sets pq_seg_ptr / pq_seg_rshift
    // (log2 of the copies per piece in LDS) -- NULL keeps the closed forms.
    int32_t pq_seg, pq_seg_rshift;
    int32_t tone_lds;           // k_pass_chain: entries of the tone table to stage in LDS (0: none)
    float pq_seg_consts[5];
    const void *pq_seg_ptr;
};

/* ---- deinterlacing (k_deinterlace.hip; reference src/shaders/deinterlacing.c) ---- */
enum plh_deint_algo {           // shaders/deinterlacing.h pl_deinterlace_algorithm
    PLH_DEINT_WEAVE = 0,
    PLH_DEINT_BOB,
    PLH_DEINT_YADIF,
    PLH_DEINT_BWDIF,
};

// The frame being deinterlaced is the sampler's `src`; rows of parity `keep` are the field being
// shown and pass through, the others are interpolated.
struct plh_deint_args {
    struct plh_view prev, next; // the neighbouring frames (`src` again where there is none)
    int32_t algo;               // enum plh_deint_algo
    int32_t keep;               // row parity of the field that is output as it is (0 = top)
    int32_t first;              // that field is the frame's first in time: the "second" temporal
                                // neighbours are (prev, cur), else (cur, next)
    int32_t intra_only;         // BWDIF without the frame it would need: spatial filter only
    int32_t skip_spatial_check; // YADIF
    float spatial_bias;         // YADIF: 1 / 255 as the reference prints it
};

// PLH_SAMPLE_DISTORT: the sampler's `pos` corners hold the canvas [-1, 1]^2 (y up); a canvas
// point p is sampled at m * p + c (texture coordinates), bilinear or by the bicubic of
// PLH_SAMPLE_BICUBIC; with an alpha mode, what lies outside the texture fades out over one texel
struct plh_distort_args {
    float m[4], c[2];           // canvas -> texture, row-major 2 x 2 + offset
    int32_t bicubic;
    int32_t alpha_mode;         // 0 = none, else enum pl_alpha_mode (premultiplied: all of rgba fades)
};

struct plh_pass {
    struct plh_sampler_args s;

    int32_t num_pre_ops;    // ops [0, num_pre_ops) run per *source texel* at tile
                            // load (fused "PASS A"), the rest per output pixel
    int32_t num_ops;
    struct plh_op ops[PLH_MAX_OPS];

    // target
    struct plh_view dst;
    int32_t base_x, base_y; // dispatch.c:1102-1124
    int32_t dir_x, dir_y;   // +1 / -1
    int32_t width, height;  // |rect|
    float out_scale[2];     // 1/width, 1/height (dispatch.c:1032-1036)
    int32_t transpose;
    int32_t frag_x0, frag_y0; // offset added to gl_FragCoord (0 for compute passes)
    // k_pass: a lane owns the 2x2 outputs [2c - pad, 2c - pad + 2) per axis; the pad that
    // makes the four bilinear footprints of a 2x upscale coincide is chosen by the host
    int32_t cell_padx, cell_pady;
    int32_t nt_store;       // streaming target stores (host: unorm targets = final frames)

    struct plh_fast_epi epi;
    struct plh_map_chain chain;

    // peak detection side output (k_peak): the 816-word measurement buffer and a zeroed scratch
    // area (PLH_PEAK_COPIES such buffers + a counter) for the kernels' partial results: k_peak_tiles
    // keeps padded accumulators and its tickets in the first 4.3 K words, k_peak_fast / k_pass_peak
    // spread their per-tile atomics over the copies and k_peak_fold adds them up; every kernel
    // leaves the area zeroed
    void *peak_buf;
    void *peak_scratch;
    // optional host mailbox (pinned, device-visible): the fold kernel also writes the 816 words
    // there and then publishes `peak_ticket` in word 816 -- the host polls that word instead of
    // waiting on the stream and copying the buffer back
    void *peak_mailbox;
    uint32_t peak_ticket;

    struct plh_deint_args deint;    // PLH_SAMPLE_DEINTERLACE
    struct plh_distort_args distort;    // PLH_SAMPLE_DISTORT
};

/* ---- error diffusion (k_errdiff.hip) ------------------------------------------ */
struct plh_errdiff_args {
    struct plh_view src, dst;
    int32_t width, height;
    int32_t quant;              // 2^depth - 1
    int32_t shift, divisor;     // pl_error_diffusion_kernel
    int32_t pattern[3][5];      // [dy][dx + 2]
    int32_t ring_rows, ring_cols;
    int32_t block_size, blocks; // one workgroup, `blocks` sequential steps
};

/* ---- overlays and blended stores (k_overlay.hip) -------------------------------- */
// What the reference draws as textured quads through the rasteriser with fixed-function blending
// (draw_overlays, src/renderer.c:811-1020; pl_dispatch_finish with blend_params): a list of
// axis-aligned parts of the target, each with an affine map from target pixel centres to the
// overlay texture, blended IN ORDER. A pixel centre p is inside [x0, x1) x [y0, y1) (the
// rasteriser's top-left rule); its texture coordinate is
//     u = u0 + ((p.x - ox) * ux + (p.y - oy) * uy),  v likewise
// (normalised; one of ux / uy is zero: the quads are only ever scaled, flipped and turned by 90
// degrees).
struct plh_overlay_part {
    float x0, y0, x1, y1;
    float ox, oy;
    float ux, uy, u0;
    float vx, vy, v0;
    float color[4];         // PLH_OVERLAY_MONOCHROME: the part's colour (osd_color)
};

enum plh_overlay_mode {
    PLH_OVERLAY_NORMAL = 0,     // color = texture(coord)
    PLH_OVERLAY_MONOCHROME,     // color = part colour; its alpha (premultiplied: all of it)
                                // times texture(coord).r after the colour ops [0, num_pre_ops)
    PLH_OVERLAY_TEXEL,          // color = texel (x - x0, y - y0): a rendered pass being blended
};

enum plh_blend_factor {         // gpu.h pl_blend_mode
    PLH_BLEND_ZERO = 0,
    PLH_BLEND_ONE,
    PLH_BLEND_SRC_ALPHA,
    PLH_BLEND_ONE_MINUS_SRC_ALPHA,
};

#define PLH_OVERLAY_TILE 16     // target pixels per tile side; one workgroup per non-empty tile
struct plh_overlay_args {
    const struct plh_overlay_part *parts;   // device
    const uint32_t *tiles;      // device: {tx | ty << 16, first, count} per non-empty tile
    const uint32_t *order;      // device: the part indices of every tile, in drawing order
    int32_t num_tiles;
    int32_t mode;               // enum plh_overlay_mode
    int32_t linear;             // overlay texture sampled with LINEAR filtering
    int32_t premultiplied;      // MONOCHROME: coverage multiplies rgba instead of a
    int32_t blend;              // 0: the colour replaces the target's
    int32_t src_rgb, dst_rgb, src_alpha, dst_alpha;     // enum plh_blend_factor
};

/* ---- launch entry points (implemented in *.hip) --------------------------- */
typedef void *plh_stream;

// returns 0 on success, a negative hipError otherwise
int plh_launch_pass(plh_stream stream, const struct plh_pass *pass);

/* ---- the two passes of a separable downscale of a one-component half-float plane as ONE launch
 * (k_lowpass2, k_ortho.hip): the contrast-recovery feature map's low-pass, reference
 * src/renderer.c:2089-2154 -> pl_shader_sample_ortho2 twice (vertical, then horizontal, through an
 * r16hf intermediate the fused kernel keeps in LDS) ---- */
struct plh_lowpass2 {
    struct plh_view src, dst;           // r16hf in, r16hf out
    float pos_v[4][2], pos_h[4][2];     // the two passes' `pos` corners (struct plh_sampler_args)
    float os_v[2], os_h[2];             // their out_scale: 1 / (width, height) of their rects
    int32_t mid_w, mid_h;               // the intermediate image: src.w x dst.h, never stored
    const float *wgt_v, *wgt_h;         // their weight tables (256 phases x stride)
    int32_t n_v, stride_v, n_h, stride_h;
    float scale_v, scale_h;
    int32_t mirror;                     // PLH_ADDRESS_MIRROR (else clamp)
    int32_t linear_trick;               // both tables hold {w0 + w1, w1 / (w0 + w1)} pairs (use_linear)
    int32_t rows_cap, cols_cap;         // the LDS tile's capacity in source rows / columns
};
// 0, a negative error, or 1 = not a shape the fused kernel takes (nothing launched)
int plh_launch_lowpass2(plh_stream stream, const struct plh_lowpass2 *args);
int plh_lowpass2_applies(const struct plh_lowpass2 *args);
int plh_launch_errdiff(plh_stream stream, const struct plh_errdiff_args *args);
// pass: s.src = the overlay texture, ops (split at num_pre_ops), dst = the target
int plh_launch_overlay(plh_stream stream, const struct plh_pass *pass,
                       const struct plh_overlay_args *args);

// POLAR phase-class setup helpers (k_polar.hip). `out` = width floats (fcoord.x
// of every output column on row 0), width ints (base texel), then height floats
// and height ints for the rows (on column 0).
int plh_launch_polar_classify(plh_stream stream, const struct plh_pass *pass, void *out);
// weights[(cy * ncx + cx) * (num_taps + 1) + t] for every tap of pass->s.taps,
// followed by the normalisation factor scale / wsum
int plh_launch_polar_weights(plh_stream stream, const struct plh_pass *pass,
                             const float *clsx, int ncx, const float *clsy, int ncy,
                             float *weights);

#ifdef __cplusplus
}
#endif

#endif // PLH_DEVICE_H_
