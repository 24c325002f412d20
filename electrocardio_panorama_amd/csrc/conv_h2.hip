// conv_h2.hip -- the grouped 1-D convolutions (K = 3, K = 7; forward and backward-data) as DIRECT convolutions whose fp32
// operands are each split into two fp16 terms and multiplied on the fp16 matrix cores with fp32 accumulation.  fp32-CLASS
// arithmetic (22..23 significant bits per operand, block-scaled by one power of two per tensor and per weight row), not IEEE fp32:
//
//     x = xh + xl + ex,  xh = fp16(x),  xl = fp16(x - xh)            |ex| <= 2^-23 |x|   (22..23 significant bits kept)
//     w = wh + wl + ew   (after a power-of-two scaling of each output row, undone in the epilogue)
//     x * w  ~  xh*wh + xh*wl + xl*wh                                (the dropped xl*wl is 2^-22 of the product)
//
// Every fp16 x fp16 product is exact in fp32 and the sums run in fp32 accumulators, so the result carries the rounding of an
// fp32 dot product plus the 2^-22..2^-23 of the operand split: measured 1.1e-7 rel-L2 on a 128-channel K = 7 layer against
// fp64 (torch's fp32 conv: 1.3e-7; the Winograd F(4,.) forms of conv_mfma.hip: 4..10e-7) -- tests/test_ops_gpu.py::test_conv_h2.
// Three `v_mfma_f32_32x32x16_f16` (16 channels each) replace eight `v_mfma_f32_32x32x2f32` per 16 channels and tap: 3 x 32
// cycles instead of 8 x 64 on a SIMD's matrix pipe (5.3x fewer; against the Winograd forms 2.5..3.4x fewer), which turns every
// conv of the train step from matrix-bound into HBM-bound.
//
// Range: fp16 holds |v| < 65504 and loses relative precision below 6.1e-5 (absolute 2^-25 below it).  Weights are scaled per
// output row at pack time (row maximum -> [2^14, 2^15)), so their split is always at full precision.  Activations and gradients
// are scaled by ONE power of two per launch, derived from the magnitude the call site's previous launch measured (x_amax ->
// [2^8, 2^9); ops.py: amax_roll, H2_HEADROOM = 64 x of growth per pass fit under it).  RANGE RESCUE (round 5): a workgroup whose
// tile nevertheless holds an element that does not fit redoes the tile with the scale its own data asks for -- the scale is divided
// out in the epilogue, so tiles of one launch may use different ones -- and no finite operand is ever clamped; what is still out of
// range afterwards is not finite and counts itself in x_clamped (the host side then skips the train step: nef_h2_taint).
// Elements more than 2^11 below the tensor's largest keep an ABSOLUTE error of <= 2^-25 / scale instead of a relative one: 2^-34 of
// the largest element at worst, below fp32's own rounding of a dot product that contains that element, but NOT a per-element
// relative bound (tests: test_conv_h2_operand_distributions).
//
// Tiling: one workgroup = 128 (TM = 2; 64 with TM = 1) output channels x 256 outputs of one sample and group, 4 waves as
// 2 (co) x 2 (t), a wave owns 32 TM x 128 = TM x 4 accumulator tiles of 32 x 32.  MFMA column n of t-tile j is output t = 4 n + j, so a lane ends up with FOUR
// ADJACENT outputs of each of its rows (16-byte stores, as the F(4,3) epilogue) and the B fragment of (tap kk, t-tile j) depends
// on s = kk + j only: a stage reads K + 3 fragment pairs from LDS instead of 4 K (K = 7: 10 instead of 28).
// B (activations): raw fp32 rows fetched through buffer descriptors a stage (16 channels) ahead, split in registers, stored to
// LDS as fp16 [plane][position][16 channels] with the positions de-interleaved by t mod 4 (fragment reads = 1 KB contiguous,
// conflict-free), double-buffered: one barrier per stage.  A (weights): never in LDS -- nef_pack_weight_h2 lays the fragments
// out in lane order, so a wave's four fragments of a (tap, 16-channel chunk) are one contiguous 4 KB read from L2, fetched a
// tap ahead.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "nefnet_hip.h"
#include "nef_common.h"

#ifndef NEF_H2_CLAMP
#define NEF_H2_CLAMP 0      // 1: clamp operands at fp16's range before the split (round 4; the range rescue makes it unnecessary)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));      // a 16-byte global load at dword alignment
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int KC = 16;            // input channels per stage = the k extent of one matrix instruction
constexpr int NTO = 256;          // outputs per workgroup
constexpr int PRO_MAX_CIN = 512;  // input channels per group the LDS table of the affine prologue holds

__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& l) {
    x0 = __builtin_amdgcn_fmed3f(x0, -65000.f, 65000.f);
    x1 = __builtin_amdgcn_fmed3f(x1, -65000.f, 65000.f);
    // h = (fp16(x0), fp16(x1));  r = x - h in ONE mixed-precision FMA per element (h * -1 + x, the fp16 source read in place:
    // bit-identical to x - float(h), two instructions fewer per element than convert-back + subtract);  l = (fp16(r0), fp16(r1))
    float r0, r1;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(x0), "v"(x1));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(x1));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l) : "v"(r0), "v"(r1));
}

#ifndef NEF_H2_SPLIT
#define NEF_H2_SPLIT 1      // 1 (round 5): the scale rides on the staging's channel-scale multiply, split by full-rate instructions
#endif
// split of an already scaled pair, no clamp (range rescue): h = (fp16(x0), fp16(x1)), l = (fp16(x0 - h0), fp16(x1 - h1)).  Four
// full-rate instructions (2.2 ns of SIMD time each, profiles/r05_valu_rates.md) where split2s spends four half-rate ones (3.75 ns).
__device__ __forceinline__ void split2n(float x0, float x1, unsigned& h, unsigned& l) {
    float r0, r1;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(x0), "v"(x1));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(x1));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l) : "v"(r0), "v"(r1));
}

// the same split with the power-of-two scale s folded into the conversions: hi = fp16(x s), lo = fp16(x s - hi), each ONE
// mixed-precision FMA per element (x s is exact, x s - hi is exact in fp32: bit-identical to split2(x0 * s, x1 * s));
// |x| is clamped at lim = 65000 / s first
__device__ __forceinline__ void split2s(float x0, float x1, float s, float lim, unsigned& h, unsigned& l) {
#if NEF_H2_CLAMP
    x0 = __builtin_amdgcn_fmed3f(x0, -lim, lim);
    x1 = __builtin_amdgcn_fmed3f(x1, -lim, lim);
#else
    (void)lim;      // no clamp: a tile whose data does not fit is redone with its own scale (range rescue), its first pass is discarded
#endif
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(h) : "v"(x0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(h) : "v"(x1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(s), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(s), "v"(h));
}

// ------------------------------------------------------------------------------------------------------------------
// Weight operand.  Logical operand of the launch: W[co][ci][kk] (forward: w[g*Cog+co][ci][kk]; transpose_flip: the
// backward-data operand, W[co' = ci][ci' = co][kk] = w[g*Cog+ci'][co'][K-1-kk]).  Packed as fp16 fragments
//     wp[g][ci'/16][kk][co'/32][plane h|l][lane = co'%32 + 32*((ci'%16)/8)][ci'%8]        (halves)
// followed by the per-row descale factors [g][co'] (floats): row co' was multiplied by 2^e, e = 14 - floor(log2(max |row|)).
// One wave per output row.
// ------------------------------------------------------------------------------------------------------------------
struct H2PackDesc {
    const float* w;
    _Float16* wp;
    int G, Cog, Cig, K, flip;      // Cog / Cig: dimensions of the LOGICAL weight tensor [G*Cog][Cig][K] that is packed
    int src_mode, src_Cr;          // nef_pack_desc.src_mode / src_Cr: how the logical tensor is read out of `w`
};

// Element (row, ci, kk) of the logical weight tensor [G*Cog][Cig][K].  src_mode 0: w itself.  src_mode 1 (round 6): the PHASE weights
// of conv1d(upsample2(x), w) formed on the fly from w [G*Cog/2][Cig][3] -- exactly poly_weights_kernel's expressions
// and row orders (src_Cr = 0: row 2 r + p; src_Cr > 0: the tile order of the polyphase forward launch), so the packed operand is
// bit-identical to packing nef_poly_weights' output and the fp32 phase tensor is never written.
__device__ __forceinline__ float h2_pack_src(const H2PackDesc& d, int64_t row, int ci, int kk) {
    if (d.src_mode == 0) return d.w[(row * d.Cig + ci) * d.K + kk];
    int64_t r;
    int p;
    if (d.src_Cr > 0) {
        const int Cr = d.src_Cr;
        const int64_t g = row / (2 * Cr);
        const int x = (int)(row - g * 2 * Cr);
        const int y = x & 127;
        p = (y >> 5) & 1;
        r = g * Cr + (x >> 7) * 64 + (y >> 6) * 32 + (y & 31);
    } else {
        r = row >> 1;
        p = (int)(row & 1);
    }
    const float* const wr = d.w + (r * d.Cig + ci) * 3;
    const float w0 = wr[0], w1 = wr[1], w2 = wr[2];
    if (p == 0) return kk == 0 ? fmaf(0.75f, w0, 0.25f * w1) : (kk == 1 ? fmaf(0.25f, w0, 0.75f * (w1 + w2)) : 0.25f * w2);
    return kk == 0 ? 0.25f * w0 : (kk == 1 ? fmaf(0.25f, w2, 0.75f * (w0 + w1)) : fmaf(0.75f, w2, 0.25f * w1));
}
constexpr int H2_PACK_MAX = 48;
struct H2PackTable { H2PackDesc d[H2_PACK_MAX]; };

__global__ __launch_bounds__(256) void pack_h2_kernel(H2PackTable tab) {
    const H2PackDesc& d = tab.d[blockIdx.y];
    const int co_n = d.flip ? d.Cig : d.Cog;       // rows / reduction channels of the launch operand
    const int ci_n = d.flip ? d.Cog : d.Cig;
    const int K = d.K;
    const int lane = threadIdx.x & 63;
    const int rows = d.G * co_n;
    const int ncot = co_n / 32, nc16 = ci_n / 16;
    float* const dsc = reinterpret_cast<float*>(d.wp + (int64_t)d.G * K * co_n * ci_n * 2);
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        const int g = row / co_n, co = row % co_n;
        float m = 0.f;
        for (int i = lane; i < ci_n * K; i += 64) {
            const int ci = i / K, kk = i % K;
            const float v = d.flip ? h2_pack_src(d, (int64_t)g * d.Cog + ci, co, K - 1 - kk) : h2_pack_src(d, (int64_t)g * d.Cog + co, ci, kk);
            m = fmaxf(m, fabsf(v));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        int e = 0;
        if (m > 0.f && m < 3e38f) {
            int ex;
            (void)frexpf(m, &ex);          // m = f * 2^ex, f in [0.5, 1)  ->  floor(log2 m) = ex - 1
            e = 14 - (ex - 1);
            e = e > 100 ? 100 : (e < -100 ? -100 : e);
        }
        const float sc = ldexpf(1.f, e);
        if (lane == 0) dsc[row] = ldexpf(1.f, -e);
        for (int i = lane; i < ci_n * K; i += 64) {
            const int ci = i / K, kk = i % K;
            const float v = sc * (d.flip ? h2_pack_src(d, (int64_t)g * d.Cog + ci, co, K - 1 - kk) : h2_pack_src(d, (int64_t)g * d.Cog + co, ci, kk));
            const _Float16 h = (_Float16)v;
            const _Float16 l = (_Float16)(v - (float)h);
            const int64_t frag = ((((int64_t)g * nc16 + ci / 16) * K + kk) * ncot + co / 32) * 2;
            const int fl = (co & 31) + 32 * ((ci & 15) >> 3);
            d.wp[(frag * 64 + fl) * 8 + (ci & 7)] = h;
            d.wp[((frag + 1) * 64 + fl) * 8 + (ci & 7)] = l;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// PRO as in conv_fwd_kernel (bit0: BatchNorm affine + ReLU of the producing layer, bit1: x2 linear upsampling of a
// half-resolution input), applied to the fp32 values before the split.
// ------------------------------------------------------------------------------------------------------------------
// A/B switches of the range rescue (tools/h2_rescue_ab.sh): RESCUE 0 = one pass, no second copy of the tile's work; AMAX 0 = the
// staging does not track the operand magnitude either (x_amax_next is then not written: timing only); EPI_ALWAYS 1 = the first
// pass runs its epilogue even when the tile has to be redone (the second pass overwrites it).
#ifndef NEF_H2_RESCUE
#define NEF_H2_RESCUE 1
#endif
#ifndef NEF_H2_AMAX
#define NEF_H2_AMAX 1
#endif
#ifndef NEF_H2_EPI_ALWAYS
#define NEF_H2_EPI_ALWAYS 0
#endif
// NEF_H2_SPLIT 1: the staging multiplies by (channel scale x launch scale) -- one product it does anyway, exact because the launch
// scale is a power of two -- and splits the scaled value; the magnitude it tracks is the scaled one (divided out once per tile).
// Bit-identical to the 0 form (scale folded into v_fma_mixlo/hi_f16).
#if NEF_H2_SPLIT
#define NEF_H2_SPLIT2(V0, V1, H, L) split2n(V0, V1, H, L)
#define NEF_H2_STAGE_SCALE xs_
#else
#define NEF_H2_SPLIT2(V0, V1, H, L) split2s(V0, V1, xs_, xlim_, H, L)
#define NEF_H2_STAGE_SCALE 1.f
#endif
#if NEF_H2_AMAX
#define NEF_H2_TRACK1(V) amax_ = fmaxf(amax_, fabsf(V));
#define NEF_H2_TRACK2(V, W) amax_ = fmaxf(amax_, fmaxf(fabsf(V), fabsf(W)));
#else
#define NEF_H2_TRACK1(V)
#define NEF_H2_TRACK2(V, W)
#endif
#ifndef NEF_H2_LAYOUT
#define NEF_H2_LAYOUT 1
#endif
#ifndef NEF_H2_T
#define NEF_H2_T 0      // timing-only builds: 1 = no activation loads in the loop, 2 = no matrix instructions, 4 = no epilogue
#endif
#ifndef NEF_H2_XORDER
#define NEF_H2_XORDER 1     // 1 (round 6): tap 1's A fragments are issued in front of the next stage's activation rows (see the stage loop)
#endif
#ifndef NEF_H2_UP_OCC3
#define NEF_H2_UP_OCC3 0      // 1: the x2-upsampling forms of the 64-channel tile at three workgroups per CU too
#endif
#ifndef NEF_H2_OCC1
#define NEF_H2_OCC1 3      // workgroups per CU the 64-channel tile is compiled for (168 VGPRs); the x2-upsampling prologue needs 2
#endif
// PACK (short rows, 8 <= T <= 64, T % 4 == 0): a tile is `tps` SAMPLES laid end to end at a pitch of T + 4 positions -- the four
// positions between two samples are staged as zeros (the zero padding of both neighbours) and their outputs are dropped; a lane's
// four adjacent outputs lie inside one sample or inside one gap.  No prologue, channel scale or statistics in this mode.
template <int K, int PRO, int TM, bool PACK = false>
__global__ __launch_bounds__(256, (TM == 1 && ((PRO & 2) == 0 || NEF_H2_UP_OCC3)) ? NEF_H2_OCC1 : 2) void conv_h2_kernel(nef_conv_args a_, int tps_, int n_tiles_, int m_tiles_) {
    constexpr int MT = 64 * TM;                    // output channels per workgroup: 2 (co) x 2 (t) waves of TM x 4 tiles
    constexpr bool UP = (PRO & 2) != 0, AFF = (PRO & 1) != 0;
    // PH (pro_mode 4, polyphase backward-data through a x2 upsampling): the input is a FULL-resolution tensor [Cin_g / 2][2 T] read as
    // Cin_g phase channels of length T -- reduction channel 2 c + p at position m is x[c][2 m + p], one 8-byte load per two channels
    constexpr bool PH = (PRO & 4) != 0;
    static_assert(!PH || (!UP && K == 3), "phase-stacked input: K = 3, no upsampling prologue");
    // PF (pro_mode 8 | affine bit, polyphase FORWARD of conv1d(upsample2(x))): x is the half-resolution tensor, staged with
    // nn.Upsample's clamped ends (position -1 = x[0], position T = x[T-1]); the launch's Cout_g rows are (channel, phase) pairs in
    // tile order -- row wm 64 + p 32 + r of a 128-row tile = phase p of channel m0 / 2 + wm 32 + r, i.e. a lane's two accumulator
    // sets are the two phases of the same channels -- and the epilogue writes y[c][2 m + p] (y: [Cout_g / 2][2 T]), eight
    // consecutive outputs per lane and row; bias and BatchNorm statistics only.
    constexpr bool PF = (PRO & 8) != 0;
    static_assert(!PF || (!UP && !PH && K == 3 && TM == 2 && !PACK), "polyphase forward: K = 3, 128-row tile");
    constexpr int NS = UP ? 2 : 1;
    constexpr int PAD = (K - 1) / 2;
    constexpr int XROW = NTO + K - 1;              // staged positions per channel: t0 - PAD .. t0 + NTO + PAD - 1
    constexpr int P4 = (XROW + 3) / 4 + 1;         // positions per (t mod 4) class
    constexpr int PLANE = 4 * P4 * 32;             // bytes of one fp16 plane of a stage
    // LDS image of a plane.  NEF_H2_LAYOUT 1 (round 5): [channel half][t mod 4][P4][8 channels] -- a 16-byte chunk per (position, half)
    // at a pitch of 16 bytes, so a fragment read (lane = position, ds_read_b128) covers 256 contiguous bytes per 16-lane group:
    // conflict-free.  0 (round 4): [t mod 4][P4][16 channels] -- 32 bytes per position, of which a lane reads one half: every
    // 16-lane group of a fragment read spans 512 bytes for 256 of data (2-way bank conflict), stores 4-way.
    constexpr int HALF = 4 * P4 * 16;
#if NEF_H2_LAYOUT
#define NEF_H2_WADDR(R, W) (((W) >> 1) * HALF + (((R) & 3) * P4 + ((R) >> 2)) * 16 + ((W) & 1) * 8)
#define NEF_H2_RPOS(S) ((((S) & 3) * P4 + ((S) >> 2)) * 16)
#else
#define NEF_H2_WADDR(R, W) ((((R) & 3) * P4 + ((R) >> 2)) * 32 + 8 * (W))
#define NEF_H2_RPOS(S) ((((S) & 3) * P4 + ((S) >> 2)) * 32)
#endif
    constexpr int NIT = (XROW + 63) / 64;
    constexpr int NSF = K + 3;                     // distinct B fragments per stage (s = tap + t-tile)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h2[];
    unsigned char* const Xl = smem_h2;             // [2 buffers][2 planes][PLANE]
    float* const Pl = reinterpret_cast<float*>(smem_h2 + 4 * PLANE);      // [2][Cin_g] prologue affine (AFF)
    float* const El = Pl + (AFF ? 2 * PRO_MAX_CIN : 0);                    // [8][MT] epilogue tables (bias, 4 x BatchNorm-backward, descale, residual scale, gate row scale)
    float* const Al = El + 8 * MT;                                         // [4] the waves' operand magnitudes of this tile (range rescue)

    // Range rescue (round 5): if an element of THIS tile turns out to exceed fp16's range under the launch's scale (the operand grew
    // more than ops.H2_HEADROOM x since the call site measured it), the workgroup does its tile again with the scale its own data
    // asks for -- the scale is a power of two that the epilogue divides out again, so tiles of one launch may use different ones.
    // Costs a wave reduction and four LDS words per tile when nothing happens; a launch no longer clamps finite data at all.
    // The tile's work is a lambda inlined TWICE (a backward branch around it made the register allocator demote the staging and
    // accumulator arrays to scratch: 1.3 .. 4.3 KB per lane): returns the tile's largest |operand| if that did not fit and the
    // epilogue was therefore skipped (`last` = false), else a negative number.
    // The two copies share NOTHING but their roots: the second one starts from opaque copies of the workgroup id, the thread id and
    // the kernel-argument pointer and derives everything again (tile, arguments, per-lane offsets).  Sharing them -- value
    // numbering finds the second copy's scalars and offsets in the first -- kept ~40 scalar and ~20 vector registers live across
    // the first copy for the sake of the second and cost the K = 3 launches 2 .. 15 % (tools/h2_rescue_ab.sh).
    typedef const nef_conv_args __attribute__((address_space(4))) kargs_t;
    auto tile_pass = [&](kargs_t* const ap, const int bid, const int tid, const int tps, const int n_tiles, const int m_tiles,
                         const float xs_force, const bool last) __attribute__((always_inline)) -> float {
#define a (*ap)
    const int tile = bid % n_tiles;
    const int gm = bid / n_tiles;
    const int mt = gm % m_tiles;
    const int g = gm / m_tiles;
    int b0, t0;
    if constexpr (PACK) {
        b0 = tile * tps;
        t0 = 0;
    } else {      // tiles of one sample 8 workgroup ids apart: same XCD, back to back (see conv_fwd_kernel)
        const int full = (n_tiles / (8 * tps)) * (8 * tps);
        if (tile < full) {
            const int grp = tile / (8 * tps), r = tile % (8 * tps);
            b0 = grp * 8 + (r & 7);
            t0 = (r >> 3) * NTO;
        } else {
            b0 = tile / tps;
            t0 = (tile - b0 * tps) * NTO;
        }
    }
    const int m0 = mt * MT;
    const int T = a.T, Cig = a.Cin_g, Cog = a.Cout_g;
    const int lane = tid & 63, wave = tid >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave_u >> 1, wn = wave_u & 1;
    const int Tin = UP ? (T >> 1) : (PH ? 2 * T : T);      // row pitch of the input tensor

    const float* const xbase = a.x + (int64_t)b0 * a.x_bs + (int64_t)g * a.x_gs;
    const __amdgpu_buffer_rsrc_t xrs = nef_rsrc(xbase);
    const _Float16* const wph = reinterpret_cast<const _Float16*>(a.wp);
    const int ncot = Cog / 32, nc16 = Cig / 16;
    // this wave's 2 TM A fragments of (chunk c, tap kk): contiguous at ((g*nc16 + c)*K + kk)*ncot*2 KB + (m0/32 + TM wm)*2 KB
    const __amdgpu_buffer_rsrc_t wrs = nef_rsrc(wph + ((int64_t)g * nc16 * K * ncot * 2 + (int64_t)(m0 / 32 + TM * wm) * 2) * 512);
    const unsigned a_tap = (unsigned)(ncot * 2 * 1024);      // bytes between taps of one chunk

    const int64_t soff = (int64_t)b0 * a.sc_bs + (int64_t)g * a.sc_gs;
    // input scale (an exact power of two, undone in the epilogue): from the magnitude this operand had at the call site's previous
    // launch (*x_amax -> [2^8, 2^9): with ops.amax_roll's follow-up rule, room for 64 x of growth since that launch before anything
    // is clamped; full precision for elements within 2^-9 of the largest), else the caller's x_scale, else 1
    float xs_ = xs_force;      // (second pass: the scale this tile's own data asks for)
    if (!last) {
        xs_ = a.x_scale != 0.f ? a.x_scale : 1.f;
        if (a.x_amax) {
            const float m_ = a.x_amax[0];
            if (m_ > 0.f && m_ < 3e38f) {
                int e_;
                (void)frexpf(m_, &e_);
                xs_ = ldexpf(1.f, 9 - e_ < 100 ? 9 - e_ : 100);
            }
        }
    }
    const int pro_row0 = AFF ? (b0 / a.pro_Bp) * a.G * Cig + g * Cig : 0;
    const unsigned avo = (unsigned)(lane * 16);
    unsigned xvo[NIT][NS];
    float lam[NIT];
    bool xok[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int r = lane + 64 * it;
        const int t = t0 + r - PAD;
        xok[it] = (r < XROW) && (t >= 0) && (t < T);
        lam[it] = 0.f;
        if constexpr (PACK) {
            const int v = r - PAD, vb = v >= 0 ? v / (T + 4) : 0, tl = v - vb * (T + 4);
            xok[it] = (r < XROW) && v >= 0 && tl < T && vb < tps && b0 + vb < a.B;
            xvo[it][0] = xok[it] ? (unsigned)((vb * (int)a.x_bs + tl) * 4) : NEF_OOB;
        } else if constexpr (UP) {
            float src = 0.5f * ((float)t + 0.5f) - 0.5f;
            if (src < 0.f) src = 0.f;
            int i0 = (int)src;
            if (i0 > Tin - 1) i0 = Tin - 1;
            const int i1 = i0 + (i0 < Tin - 1 ? 1 : 0);
            lam[it] = src - (float)i0;
            xvo[it][0] = xok[it] ? (unsigned)(i0 * 4) : NEF_OOB;
            xvo[it][NS - 1] = xok[it] ? (unsigned)(i1 * 4) : NEF_OOB;
        } else if constexpr (PF) {
            xok[it] = (r < XROW) && (t >= -1) && (t <= T);
            const int tc = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
            xvo[it][0] = xok[it] ? (unsigned)(tc * 4) : NEF_OOB;
        } else {
            xvo[it][0] = xok[it] ? (unsigned)(t * (PH ? 8 : 4)) : NEF_OOB;
        }
    }
    // x2-upsampling prologue: a lane stages INTERVALS of the half-resolution row instead of positions -- interval m = t0/2 - 1 + u
    // (u = lane + 64 it, it = 0, 1) holds the sources x[m], x[m + 1] of the two outputs t = 2m + 1 (position r = 2u: 0.75 / 0.25) and
    // t = 2m + 2 (r = 2u + 1: 0.25 / 0.75; output 0 is x[0] itself: weights 0 / 1 on the pair (x[-1] -> 0, x[0])), exactly
    // nn.Upsample's values (and the (1 - lam) a + lam b expression of the position form): two loads per two outputs instead of
    // four, the affine + ReLU once per source.  The right halo interval u = 128 (r = 256, 257) is staged by lanes 0..3 of each
    // wave, one of the wave's four channels each.
    unsigned uvo[2][2], mvo[2];
    bool uok[2][2], mok[2];
    float ulam[2];
    if constexpr (UP) {
        static_assert(!UP || (K == 3 && NTO == 256), "interval staging: K = 3, 256-column tiles");
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int u = it < 2 ? lane + 64 * it : 128;
            const int m = (t0 >> 1) - 1 + u;
            const bool ok0 = 2 * m + 1 >= 0 && 2 * m + 1 < T, ok1 = 2 * m + 2 >= 0 && 2 * m + 2 < T;
            const bool any = ok0 || ok1;
            const int mb = m + 1 < Tin ? m + 1 : Tin - 1;
            const unsigned oa = (any && m >= 0 && m < Tin) ? (unsigned)(m * 4) : NEF_OOB;
            const unsigned ob = (any && m < Tin) ? (unsigned)(mb * 4) : NEF_OOB;
            if (it < 2) {
                uok[it][0] = ok0, uok[it][1] = ok1;
                uvo[it][0] = oa, uvo[it][1] = ob;
                ulam[it] = m == -1 ? 1.f : 0.75f;
            } else {      // lane l < 4: channel 4 wave + l (its row starts l Tin floats behind the wave's first)
                mok[0] = ok0 && lane < 4, mok[1] = ok1 && lane < 4;
                mvo[0] = (lane < 4 && oa != NEF_OOB) ? oa + (unsigned)(lane * Tin * 4) : NEF_OOB;
                mvo[1] = (lane < 4 && ob != NEF_OOB) ? ob + (unsigned)(lane * Tin * 4) : NEF_OOB;
            }
        }
    }
    const float xlim_ = 65000.f / xs_;
    (void)xlim_;
    float amax_ = 0.f;
    float over_ret = -1.f;
    (void)over_ret;
    f32x16 acc[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- operand streams
    float xreg[4][UP ? 2 : NIT][NS];          // this wave's 4 channels (4 wave + 0..3) of the stage in flight
    float xm[2];                              // UP: the right halo interval, channel 4 wave + lane (lanes 0..3)
#define NEF_H2X_ISSUE(C0, RS)                                                                                        \
    {                                                                                                               \
        _Pragma("unroll") for (int rr = 0; rr < 4; ++rr) {                                                          \
            const unsigned so = (unsigned)(((C0) + 4 * wave_u + rr) * Tin * 4);                                     \
            if constexpr (UP) {                                                                                     \
                _Pragma("unroll") for (int it = 0; it < 2; ++it)                                                    \
                    _Pragma("unroll") for (int ns = 0; ns < 2; ++ns) xreg[rr][it][ns] = nef_buf_f32(RS, uvo[it][ns], so); \
            } else if constexpr (PH) {                                                                              \
                const unsigned so2 = (unsigned)((((C0) >> 1) + 2 * wave_u + (rr >> 1)) * Tin * 4 + (rr & 1) * 4);   \
                _Pragma("unroll") for (int it = 0; it < NIT; ++it) xreg[rr][it][0] = nef_buf_f32(RS, xvo[it][0], so2); \
            } else {                                                                                                \
                _Pragma("unroll") for (int it = 0; it < NIT; ++it) xreg[rr][it][0] = nef_buf_f32(RS, xvo[it][0], so); \
            }                                                                                                       \
        }                                                                                                           \
        if constexpr (UP) {                                                                                         \
            const unsigned so = (unsigned)(((C0) + 4 * wave_u) * Tin * 4);                                          \
            xm[0] = nef_buf_f32(RS, mvo[0], so);                                                                    \
            xm[1] = nef_buf_f32(RS, mvo[1], so);                                                                    \
        }                                                                                                           \
    }
    // split the stage in registers and store it: position r of channel c -> plane[(r & 3) * P4 + (r >> 2)][c]
#define NEF_H2X_STORE(C0, BUFP)                                                                                      \
    {                                                                                                               \
        float sa_[4], pa_[4], pb_[4];                                                                               \
        _Pragma("unroll") for (int rr = 0; rr < 4; ++rr) {                                                          \
            sa_[rr] = (a.in_scale ? a.in_scale[soff + (C0) + 4 * wave + rr] : 1.f) * NEF_H2_STAGE_SCALE;            \
            pa_[rr] = 1.f, pb_[rr] = 0.f;                                                                           \
            if constexpr (AFF) {                                                                                    \
                pa_[rr] = Pl[(C0) + 4 * wave_u + rr];                                                               \
                pb_[rr] = Pl[Cig + (C0) + 4 * wave_u + rr];                                                         \
            }                                                                                                       \
        }                                                                                                           \
        if constexpr (UP) {                                                                                         \
            _Pragma("unroll") for (int it = 0; it < 2; ++it) {                                                      \
                float v0_[4], v1_[4];                                                                               \
                _Pragma("unroll") for (int rr = 0; rr < 4; ++rr) {                                                  \
                    float sa = xreg[rr][it][0], sb = xreg[rr][it][1];                                               \
                    if constexpr (AFF) sa = fmaxf(fmaf(sa, pa_[rr], pb_[rr]), 0.f), sb = fmaxf(fmaf(sb, pa_[rr], pb_[rr]), 0.f); \
                    float o0 = (1.f - 0.25f) * sa + 0.25f * sb;                                                     \
                    float o1 = (1.f - ulam[it]) * sa + ulam[it] * sb;                                               \
                    o0 = uok[it][0] ? o0 : 0.f;                                                                     \
                    o1 = uok[it][1] ? o1 : 0.f;                                                                     \
                    o0 *= sa_[rr], o1 *= sa_[rr];                                                                   \
                    NEF_H2_TRACK2(o0, o1)                                              \
                    v0_[rr] = o0, v1_[rr] = o1;                                                                     \
                }                                                                                                   \
                const int r = 2 * (lane + 64 * it);                                                                 \
                unsigned h0_, l0_, h1_, l1_, h2_, l2_, h3_, l3_;                                                    \
                NEF_H2_SPLIT2(v0_[0], v0_[1], h0_, l0_);                                                      \
                NEF_H2_SPLIT2(v0_[2], v0_[3], h1_, l1_);                                                      \
                NEF_H2_SPLIT2(v1_[0], v1_[1], h2_, l2_);                                                      \
                NEF_H2_SPLIT2(v1_[2], v1_[3], h3_, l3_);                                                      \
                unsigned char* p0_ = (BUFP) + NEF_H2_WADDR(r, wave);                                                \
                unsigned char* p1_ = (BUFP) + NEF_H2_WADDR(r + 1, wave);                                            \
                *reinterpret_cast<u32x2*>(p0_) = u32x2{h0_, h1_};                                                   \
                *reinterpret_cast<u32x2*>(p0_ + PLANE) = u32x2{l0_, l1_};                                           \
                *reinterpret_cast<u32x2*>(p1_) = u32x2{h2_, h3_};                                                   \
                *reinterpret_cast<u32x2*>(p1_ + PLANE) = u32x2{l2_, l3_};                                           \
            }                                                                                                       \
            if (lane < 4) {      /* the right halo interval: positions 256, 257 of channel 4 wave + lane */          \
                float sa = xm[0], sb = xm[1];                                                                       \
                if constexpr (AFF) {                                                                                \
                    const float pa = Pl[(C0) + 4 * wave_u + lane], pb = Pl[Cig + (C0) + 4 * wave_u + lane];         \
                    sa = fmaxf(fmaf(sa, pa, pb), 0.f), sb = fmaxf(fmaf(sb, pa, pb), 0.f);                           \
                }                                                                                                   \
                const float sc = (a.in_scale ? a.in_scale[soff + (C0) + 4 * wave + lane] : 1.f) * NEF_H2_STAGE_SCALE; \
                float o0 = (1.f - 0.25f) * sa + 0.25f * sb;                                                         \
                float o1 = (1.f - 0.75f) * sa + 0.75f * sb;                                                         \
                o0 = mok[0] ? o0 * sc : 0.f;                                                                        \
                o1 = mok[1] ? o1 * sc : 0.f;                                                                        \
                NEF_H2_TRACK2(o0, o1)                                                  \
                unsigned h_, l_;                                                                                    \
                NEF_H2_SPLIT2(o0, o1, h_, l_);                                                                \
                unsigned char* p0_ = (BUFP) + NEF_H2_WADDR(256, wave) + 2 * lane;                                   \
                unsigned char* p1_ = (BUFP) + NEF_H2_WADDR(257, wave) + 2 * lane;                                   \
                *reinterpret_cast<unsigned short*>(p0_) = (unsigned short)(h_ & 0xffffu);                           \
                *reinterpret_cast<unsigned short*>(p0_ + PLANE) = (unsigned short)(l_ & 0xffffu);                   \
                *reinterpret_cast<unsigned short*>(p1_) = (unsigned short)(h_ >> 16);                               \
                *reinterpret_cast<unsigned short*>(p1_ + PLANE) = (unsigned short)(l_ >> 16);                       \
            }                                                                                                       \
        } else                                                                                                      \
        _Pragma("unroll") for (int it = 0; it < NIT; ++it) {                                                        \
            float v_[4];                                                                                            \
            _Pragma("unroll") for (int rr = 0; rr < 4; ++rr) {                                                      \
                float v = xreg[rr][it][0];                                                                          \
                if constexpr (AFF) v = fmaxf(fmaf(v, pa_[rr], pb_[rr]), 0.f);                                       \
                if constexpr (AFF) v = xok[it] ? v : 0.f;                                                           \
                v *= sa_[rr];                                                                                       \
                NEF_H2_TRACK1(v)                                                                                    \
                v_[rr] = v;                                                                                         \
            }                                                                                                       \
            const int r = lane + 64 * it;                                                                           \
            unsigned h0_, l0_, h1_, l1_;                                                                            \
            NEF_H2_SPLIT2(v_[0], v_[1], h0_, l0_);                                                            \
            NEF_H2_SPLIT2(v_[2], v_[3], h1_, l1_);                                                            \
            const u32x2 hv = {h0_, h1_}, lv = {l0_, l1_};                                                           \
            if (r < XROW) {                                                                                         \
                unsigned char* p_ = (BUFP) + NEF_H2_WADDR(r, wave);                                                 \
                *reinterpret_cast<u32x2*>(p_) = hv;                                                                 \
                *reinterpret_cast<u32x2*>(p_ + PLANE) = lv;                                                         \
            }                                                                                                       \
        }                                                                                                           \
    }
    h16x8 fa[2][2 * TM];             // [set][2 * co-tile + plane]
#define NEF_H2A_ISSUE(CH, KK, SET)                                                                                   \
    {                                                                                                               \
        const unsigned so_ = (unsigned)(((CH) * K + (KK)) * a_tap);                                                 \
        _Pragma("unroll") for (int q = 0; q < 2 * TM; ++q)                                                          \
            fa[SET][q] = __builtin_bit_cast(h16x8, nef_buf_f32x4(wrs, avo, so_ + (unsigned)(q * 1024)));            \
    }

    NEF_H2X_ISSUE(0, xrs)
    NEF_H2A_ISSUE(0, 0, 0)
    if (tid < MT) {     // epilogue tables, published by the barrier behind the first stage's LDS stores
        const int ch_ = g * Cog + m0 + (int)tid;
        const float* const dsc = reinterpret_cast<const float*>(wph + (int64_t)a.G * K * Cog * Cig * 2);
        if constexpr (PF) El[tid] = a.bias ? a.bias[g * (Cog >> 1) + (m0 >> 1) + ((int)tid >> 6) * 32 + ((int)tid & 31)] : 0.f;
        else
        El[tid] = a.bias ? a.bias[ch_] : 0.f;
        El[5 * MT + tid] = dsc[ch_] / xs_;
        El[6 * MT + tid] = (a.res_scale && b0 < a.B) ? a.res_scale[(int64_t)b0 * a.rs_bs + (int64_t)g * a.rs_gs + m0 + (int)tid] : 1.f;
        El[7 * MT + tid] = (a.gate_rowscale && b0 < a.B) ? a.gate_rowscale[(int64_t)b0 * a.gr_bs + (int64_t)g * a.gr_gs + m0 + (int)tid] : 1.f;
        if (a.bnb_slots) {
            const int pr_ = (b0 / a.bnb_Bp) * a.G * Cog + ch_;
            El[MT + tid] = a.bnb_mean[pr_];
            El[2 * MT + tid] = a.bnb_invstd[pr_];
            El[3 * MT + tid] = a.bnb_a[pr_];
            El[4 * MT + tid] = a.bnb_b[pr_];
        }
    }
    if constexpr (AFF) {
        for (int i = tid; i < Cig; i += 256) {
            Pl[i] = a.pro_a[pro_row0 + i];
            Pl[Cig + i] = a.pro_b[pro_row0 + i];
        }
        __syncthreads();
    }
    NEF_H2X_STORE(0, Xl)
    __syncthreads();

    const int nst = Cig / KC;
    // this lane's fragment address inside a plane: position wn * 32 + lo (+ the s-dependent constant), channels 8 hi ..
    const unsigned fb_lane = NEF_H2_LAYOUT ? (unsigned)(hi * HALF + (wn * 32 + lo) * 16) : (unsigned)((wn * 32 + lo) * 32 + hi * 16);
    for (int st = 0; st < nst; ++st) {
        const unsigned char* const xb = Xl + (st & 1) * (2 * PLANE) + fb_lane;
        const bool more = st + 1 < nst;
        const __amdgpu_buffer_rsrc_t xrs_n = nef_rsrc_n(xbase, more ? 0x7FFFFFFCu : 0u);      // branch-free: see conv_wino4_kernel
        // Issue order of the stage's vector-memory loads (round 6, NEF_H2_XORDER 1).  Loads return IN ORDER, so the wait for an A
        // fragment also waits for every load issued before it.  Rounds 4-5 issued the next stage's activation rows first and tap 1's A
        // fragments behind them: the wait in front of tap 1 then covered the activation rows -- they had ONE tap (768 cycles at K = 3,
        // 128-row tile) to arrive from HBM, whatever the length of the stage.  Now tap 1's A fragments go out first: the first wait that
        // covers the activation rows is the one in front of tap 2, two taps after their issue.
#if !(NEF_H2_T & 1) && !NEF_H2_XORDER
        NEF_H2X_ISSUE((st + 1) * KC, xrs_n)
#endif
        h16x8 fb[5][2];              // ring over s: [slot][plane]
#define NEF_H2B_LOAD(S)                                                                                              \
    {                                                                                                               \
        const unsigned char* p_ = xb + NEF_H2_RPOS(S);                                                              \
        fb[(S) % 5][0] = *reinterpret_cast<const h16x8*>(p_);                                                       \
        fb[(S) % 5][1] = *reinterpret_cast<const h16x8*>(p_ + PLANE);                                               \
    }
        NEF_H2B_LOAD(0)
        NEF_H2B_LOAD(1)
        NEF_H2B_LOAD(2)
        NEF_H2B_LOAD(3)
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            // next tap's A fragments (the next stage's first tap behind the last one; past the end: a repeat, harmless)
            if (kk + 1 < K) NEF_H2A_ISSUE(st, kk + 1, (kk + 1) & 1)
            else NEF_H2A_ISSUE(more ? st + 1 : st, 0, (kk + 1) & 1)
#if !(NEF_H2_T & 1) && NEF_H2_XORDER
            if (kk == 0) {
                __builtin_amdgcn_sched_barrier(0);      // keep the A issue in front of the activation rows
                NEF_H2X_ISSUE((st + 1) * KC, xrs_n)
            }
#endif
            if (kk + 4 < NSF) NEF_H2B_LOAD(kk + 4)
            __builtin_amdgcn_s_setprio(1);      // scheduling fence (see conv_wino_kernel)
            // product-major order: the three MFMAs that accumulate into one tile are 4 TM instructions apart (never back to
            // back on the same accumulator)
#if !(NEF_H2_T & 2)
            const int s_ = kk & 1;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s_][2 * i], fb[(kk + j) % 5][0], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s_][2 * i], fb[(kk + j) % 5][1], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s_][2 * i + 1], fb[(kk + j) % 5][0], acc[i][j], 0, 0, 0);
#endif
        }
#undef NEF_H2B_LOAD
        // K odd: the set toggles K times per stage, so stage st + 1 finds its tap 0 in set (K & 1) ^ ... -- keep it simple:
        // the last issue above wrote set (K & 1); with K odd that is set 1, but tap 0 of the next stage reads set 0
        if constexpr ((K & 1) != 0) {
#pragma unroll
            for (int q = 0; q < 2 * TM; ++q) fa[0][q] = fa[1][q];
        }
        // (round 6, measured and not kept: the split + LDS stores in front of the LAST tap's matrix instructions instead of behind them --
        // the compiler keeps the two groups apart, and the launches read +-1 %: profiles/r06_xorder_ab.md)
        if (more) NEF_H2X_STORE((st + 1) * KC, Xl + ((st + 1) & 1) * (2 * PLANE))
        else {      // every element of the tile has been staged: publish this wave's magnitude with the loop's last barrier
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) amax_ = fmaxf(amax_, __shfl_xor(amax_, o, 64));
#if NEF_H2_SPLIT
            amax_ *= 1.f / xs_;      // tracked on the scaled values (exact: a power of two)
#endif
            if (lane == 0) Al[wave_u] = amax_;
        }
        __syncthreads();
    }
    {
        const float wg_ = fmaxf(fmaxf(Al[0], Al[1]), fmaxf(Al[2], Al[3]));      // workgroup-uniform
        if (NEF_H2_RESCUE && !last && !(wg_ * xs_ < 65000.f) && wg_ < 3e38f) {
#if NEF_H2_EPI_ALWAYS
            over_ret = wg_;
#else
            __syncthreads();      // (Al is rewritten by the second pass)
            return wg_;
#endif
        }
    }
#undef NEF_H2X_ISSUE
#undef NEF_H2X_STORE
#undef NEF_H2A_ISSUE
#undef NEF_H2_WADDR
#undef NEF_H2_RPOS

    // (amax_ is wave-uniform here: reduced in the last stage)  what is still out of range after the rescue is not finite
    if (a.x_clamped && lane == 0 && !(amax_ * xs_ < 65000.f)) atomicAdd(a.x_clamped, 1);
    if (a.x_amax_next) {
        if (lane == 0 && amax_ < 3e38f) {
            unsigned* const p_ = reinterpret_cast<unsigned*>(a.x_amax_next);
            const unsigned b_ = __builtin_bit_cast(unsigned, amax_);       // non-negative floats order like their bit patterns
            if (b_ > __atomic_load_n(p_, __ATOMIC_RELAXED)) atomicMax(p_, b_);
        }
    }
#if NEF_H2_T & 4
    {
        float z_ = 0.f;
        for (int i = 0; i < TM; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) z_ += acc[i][j][r];
        if (z_ == 12345.678f) a.y[0] = z_;
        return -1.f;
    }
#endif
    if constexpr (PF) {
        // ---- polyphase epilogue: the lane's four adjacent half-resolution columns m .. m + 3 of both phases = outputs 2 m .. 2 m + 7
        const int t = t0 + wn * 128 + 4 * lo;
        const bool inb = b0 < a.B;
        const bool live0 = inb && t < T, live1 = inb && t + 2 < T;
        const int Cr = Cog >> 1;
        const int cw = (m0 >> 1) + wm * 32;             // first channel of this wave's rows
        const int erow = wm * 64 + 4 * hi;
        const __amdgpu_buffer_rsrc_t yrs = nef_rsrc(a.y + (int64_t)b0 * a.y_bs + (int64_t)g * a.y_gs + (int64_t)cw * (2 * T));
        const unsigned yv0 = live0 ? (unsigned)((4 * hi * 2 * T + 2 * t) * 4) : NEF_OOB;
        const unsigned yv1 = live1 ? (unsigned)((4 * hi * 2 * T + 2 * t + 4) * 4) : NEF_OOB;
        float* const slot_out = a.stats;
        float sv[32];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int row = ((q + 8 * h) & 3) + 8 * ((q + 8 * h) >> 2);
                const float ds0 = El[5 * MT + erow + row], ds1 = El[5 * MT + erow + 32 + row], bv = El[erow + row];
                float y0[4], y1[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y0[e] = fmaf(acc[0][e][q + 8 * h], ds0, bv);
                    y1[e] = fmaf(acc[TM - 1][e][q + 8 * h], ds1, bv);
                }
                const f32x4 o0 = {y0[0], y1[0], y0[1], y1[1]}, o1 = {y0[2], y1[2], y0[3], y1[3]};
                nef_buf_store_f32x4(o0, yrs, yv0, (unsigned)(row * 2 * T * 4));
                nef_buf_store_f32x4(o1, yrs, yv1, (unsigned)(row * 2 * T * 4));
                if (slot_out) {
                    const float s0 = live0 ? (y0[0] + y1[0]) + (y0[1] + y1[1]) : 0.f, s1 = live1 ? (y0[2] + y1[2]) + (y0[3] + y1[3]) : 0.f;
                    const float q0 = live0 ? fmaf(y0[0], y0[0], y1[0] * y1[0]) + fmaf(y0[1], y0[1], y1[1] * y1[1]) : 0.f;
                    const float q1 = live1 ? fmaf(y0[2], y0[2], y1[2] * y1[2]) + fmaf(y0[3], y0[3], y1[3] * y1[3]) : 0.f;
                    sv[2 * (q + 8 * h)] = s0 + s1;
                    sv[2 * (q + 8 * h) + 1] = q0 + q1;
                }
            }
        }
        if (slot_out) {      // (the butterfly of the other forms)
#pragma unroll
            for (int step = 0; step < 5; ++step) {
                const int off = 16 >> step;
                const bool up = (lo & off) != 0;
#pragma unroll
                for (int k = 0; k < off; ++k) {
                    const float send = up ? sv[k] : sv[k + off];
                    const float keep = up ? sv[k + off] : sv[k];
                    sv[k] = keep + __shfl_xor(send, off, 64);
                }
            }
            const int r = lo >> 1;
            const int ch = g * Cr + cw + 4 * hi + (r & 3) + 8 * (r >> 2);
            const int64_t nslot = (int64_t)tps * 2;
            const int64_t slot = (int64_t)b0 * nslot + (int64_t)(t0 / NTO) * 2 + wn;
            if (inb) slot_out[((int64_t)ch * a.B * nslot + slot) * 2 + (lo & 1)] = sv[0];
        }
#if NEF_H2_EPI_ALWAYS
        __syncthreads();
        return over_ret;
#else
        return -1.f;
#endif
    }
    // ---- epilogue: descale, then bias / residual / ReLU / dropout / gate on the four adjacent outputs a lane owns per row
    const int64_t ctot = (int64_t)a.G * Cog;
    int t = t0 + wn * 128 + 4 * lo;
    int bq = b0;                                         // the sample of this lane's outputs
    bool inb = b0 < a.B;
    if constexpr (PACK) {
        const int vp = wn * 128 + 4 * lo, vb = vp / (T + 4);
        t = vp - vb * (T + 4);
        bq = b0 + vb;
        inb = vb < tps && bq < a.B && t < T;
    }
    const int vbq = bq - b0;
    // this lane's sample offset in dense [b][c][t] tensors (elements); lanes without an output stay on the tile's first sample
    const int vb_ct = (PACK && inb) ? vbq * (int)(ctot * T) : 0;
    const bool live[2] = {inb && t < T, inb && t + 2 < T};
    const int ts[2] = {live[0] ? t : 0, live[1] ? t + 2 : 0};
    const bool ragged = PACK ? false : t0 + NTO > T;     // workgroup-uniform
    float* const slot_out = a.bnb_slots ? a.bnb_slots : a.stats;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int cobase = m0 + wm * (32 * TM) + i * 32 + 4 * hi;
        const int erow0 = wm * (32 * TM) + i * 32 + 4 * hi;
        float sv[32];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#define NEF_ROW(q) ((((q) + 8 * h) & 3) + 8 * (((q) + 8 * h) >> 2))
            float y[8][4];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
#if defined(NEF_H2_DBG) && NEF_H2_DBG == 1
                const float ds = reinterpret_cast<const float*>(wph + (int64_t)a.G * K * Cog * Cig * 2)[g * Cog + cobase + NEF_ROW(q)] / xs_;
                const float bv = a.bias ? a.bias[g * Cog + cobase + NEF_ROW(q)] : 0.f;
#elif defined(NEF_H2_DBG) && NEF_H2_DBG == 2
                const float ds = 1.f, bv = 1000.f;
#else
                const float ds = El[5 * MT + erow0 + NEF_ROW(q)];
                const float bv = El[erow0 + NEF_ROW(q)];
#endif
#pragma unroll
                for (int e = 0; e < 4; ++e) y[q][e] = fmaf(acc[i][e][q + 8 * h], ds, bv);
            }
#define NEF_EPI_FETCH4(PTR, BS, GS, DST)                                                                              \
    if (!ragged) {                                                                                                  \
        const __amdgpu_buffer_rsrc_t rs_ =                                                                          \
            nef_rsrc((PTR) + (int64_t)b0 * (BS) + (int64_t)g * (GS) + (int64_t)(m0 + wm * (32 * TM) + i * 32) * T);        \
        const unsigned vo_ = inb ? (unsigned)((4 * hi * T + t) * 4) + (PACK ? (unsigned)(vbq * (int)(BS) * 4) : 0u) : NEF_OOB; \
        _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                             \
            const f32x4 t4 = nef_buf_f32x4(rs_, vo_, (unsigned)(NEF_ROW(q) * T * 4));                               \
            DST[q][0] = t4[0]; DST[q][1] = t4[1]; DST[q][2] = t4[2]; DST[q][3] = t4[3];                             \
        }                                                                                                           \
    } else {                                                                                                        \
        _Pragma("unroll") for (int pr = 0; pr < 2; ++pr) {                                                          \
            const float* p_ = (PTR) + (int64_t)b0 * (BS) + (int64_t)g * (GS) + (int64_t)cobase * T + ts[pr];        \
            _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                         \
                const f32x2 t2 = *reinterpret_cast<const f32x2*>(p_ + (int64_t)NEF_ROW(q) * T);                     \
                DST[q][2 * pr] = t2[0];                                                                             \
                DST[q][2 * pr + 1] = t2[1];                                                                         \
            }                                                                                                       \
        }                                                                                                           \
    }
            if (a.res) {
                float rv[8][4];
                NEF_EPI_FETCH4(a.res, a.res_bs, a.res_gs, rv)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float rs_ = El[6 * MT + erow0 + NEF_ROW(q)];      // (1 without res_scale: the same sum bit for bit)
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[q][e] = fmaf(rv[q][e], rs_, y[q][e]);
                }
            }
            if (a.relu) {
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[q][e] = fmaxf(y[q][e], 0.f);
            }
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {        // dropout works on the two output pairs (t, t+1), (t+2, t+3)
                if (a.mask) {
                    const uint8_t* mp = a.mask + ((int64_t)b0 * ctot + (int64_t)g * Cog + cobase) * T + ts[pr] + vb_ct;
                    unsigned short t8[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) t8[q] = *reinterpret_cast<const unsigned short*>(mp + (int64_t)NEF_ROW(q) * T);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        y[q][2 * pr] *= (float)(t8[q] & 0xff) * a.drop_scale;
                        y[q][2 * pr + 1] *= (float)(t8[q] >> 8) * a.drop_scale;
                    }
                } else if (a.drop_p > 0.f) {
                    const int64_t d0 = ((int64_t)b0 * ctot + (int64_t)g * Cog + cobase) * T + ts[pr] + vb_ct;
                    const uint64_t seed = a.rng_seed + (a.rng_seed_dev ? a.rng_seed_dev[0] : 0ull);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const uint64_t dense = (uint64_t)(d0 + (int64_t)NEF_ROW(q) * T);
                        float u0, u1;
                        nef_rng_uniform2(seed, dense, u0, u1);
                        y[q][2 * pr] = (u0 >= a.drop_p) ? y[q][2 * pr] * a.drop_scale : 0.f;
                        y[q][2 * pr + 1] = (u1 >= a.drop_p) ? y[q][2 * pr + 1] * a.drop_scale : 0.f;
                    }
                }
            }
            if (a.gate) {
                float gv[8][4];
                NEF_EPI_FETCH4(a.gate, a.gate_bs, a.gate_gs, gv)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (a.stats_mode == 1) {      // sum_t (ungated output) x gate: the channel scaling's own gradient (nef_chscale_bwd's gs)
                        const float d0 = live[0] ? fmaf(y[q][0], gv[q][0], y[q][1] * gv[q][1]) : 0.f;
                        const float d1 = live[1] ? fmaf(y[q][2], gv[q][2], y[q][3] * gv[q][3]) : 0.f;
                        sv[2 * (q + 8 * h)] = d0 + d1;
                        sv[2 * (q + 8 * h) + 1] = 0.f;
                    }
                    const float gs_ = a.gate_scale * El[7 * MT + erow0 + NEF_ROW(q)];      // (row scale 1 without gate_rowscale)
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[q][e] = gv[q][e] > 0.f ? y[q][e] * gs_ : 0.f;
                }
            }
#undef NEF_EPI_FETCH4
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                if (live[pr] && !live[1]) {      // the half-live quad at the end of a row with T % 4 == 2
                    float* yp = a.y + (int64_t)b0 * a.y_bs + (int64_t)g * a.y_gs + (int64_t)cobase * T + t + 2 * pr;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        f32x2 o;
                        o[0] = y[q][2 * pr];
                        o[1] = y[q][2 * pr + 1];
                        *reinterpret_cast<f32x2*>(yp + (int64_t)NEF_ROW(q) * T) = o;
                    }
                }
            }
            {
                const __amdgpu_buffer_rsrc_t yrs =
                    nef_rsrc(a.y + (int64_t)b0 * a.y_bs + (int64_t)g * a.y_gs + (int64_t)(m0 + wm * (32 * TM) + i * 32) * T);
                const unsigned yvo = live[1] ? (unsigned)((4 * hi * T + t) * 4) + (PACK ? (unsigned)(vbq * (int)a.y_bs * 4) : 0u) : NEF_OOB;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    f32x4 o;
                    o[0] = y[q][0];
                    o[1] = y[q][1];
                    o[2] = y[q][2];
                    o[3] = y[q][3];
                    nef_buf_store_f32x4(o, yrs, yvo, (unsigned)(NEF_ROW(q) * T * 4));
                }
            }
            if (a.bnb_slots && a.bnb_up) {      // see conv_wino4_kernel: BatchNorm-backward sums through the x2 upsampling's adjoint
                const int Lh = T >> 1;
                const float* xp = a.bnb_x + ((int64_t)b0 * ctot + (int64_t)g * Cog + cobase) * Lh;
                const int j2 = live[0] ? (t >> 1) : 0;
                const int im1 = j2 > 0 ? j2 - 1 : 0, i1 = j2 + 1 < Lh ? j2 + 1 : Lh - 1, ip2 = j2 + 2 < Lh ? j2 + 2 : Lh - 1;
                // the four half-resolution samples j2 - 1 .. j2 + 2 of a row: one 16-byte load away from the row ends (every lane
                // but the first and the last of a row), four clamped scalar loads there
                const bool interior = j2 >= 1 && j2 + 2 < Lh;
                auto sums = [&](int q, float xa, float xb_, float xc, float xd) __attribute__((always_inline)) {
                    const int er = erow0 + NEF_ROW(q);
                    const float af = El[3 * MT + er], bf = El[4 * MT + er];
                    const float mf = El[MT + er], is = El[2 * MT + er];
                    const float ma = fmaf(xa, af, bf) > 0.f ? 1.f : 0.f, mb = fmaf(xb_, af, bf) > 0.f ? 1.f : 0.f;
                    const float mc = fmaf(xc, af, bf) > 0.f ? 1.f : 0.f, md = fmaf(xd, af, bf) > 0.f ? 1.f : 0.f;
                    const float ha = ma * ((xa - mf) * is), hb = mb * ((xb_ - mf) * is);
                    const float hc = mc * ((xc - mf) * is), hd = md * ((xd - mf) * is);
                    const float g0 = live[0] ? y[q][0] : 0.f, g1 = live[0] ? y[q][1] : 0.f;
                    const float g2 = live[1] ? y[q][2] : 0.f, g3 = live[1] ? y[q][3] : 0.f;
                    sv[2 * (q + 8 * h)] = fmaf(g0, fmaf(0.75f, mb, 0.25f * ma), g1 * fmaf(0.75f, mb, 0.25f * mc)) +
                                          fmaf(g2, fmaf(0.75f, mc, 0.25f * mb), g3 * fmaf(0.75f, mc, 0.25f * md));
                    sv[2 * (q + 8 * h) + 1] = fmaf(g0, fmaf(0.75f, hb, 0.25f * ha), g1 * fmaf(0.75f, hb, 0.25f * hc)) +
                                              fmaf(g2, fmaf(0.75f, hc, 0.25f * hb), g3 * fmaf(0.75f, hc, 0.25f * hd));
                };
                if (interior) {
                    f32x4_a4 xv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) xv[q] = *reinterpret_cast<const f32x4_a4*>(xp + (int64_t)NEF_ROW(q) * Lh + (j2 - 1));
#pragma unroll
                    for (int q = 0; q < 8; ++q) sums(q, xv[q][0], xv[q][1], xv[q][2], xv[q][3]);
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float* xr = xp + (int64_t)NEF_ROW(q) * Lh;
                        sums(q, xr[im1], xr[j2], xr[i1], xr[ip2]);
                    }
                }
            } else if (a.bnb_slots) {
                const float* xp = a.bnb_x + ((int64_t)b0 * ctot + (int64_t)g * Cog + cobase) * T;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int row = NEF_ROW(q);
                    const int er = erow0 + row;
                    const float af = El[3 * MT + er], bf = El[4 * MT + er];
                    const float mf = El[MT + er], is = El[2 * MT + er];
                    const f32x2 x01 = *reinterpret_cast<const f32x2*>(xp + (int64_t)row * T + ts[0]);
                    const f32x2 x23 = *reinterpret_cast<const f32x2*>(xp + (int64_t)row * T + ts[1]);
                    const float g0 = (live[0] && fmaf(x01[0], af, bf) > 0.f) ? y[q][0] : 0.f;
                    const float g1 = (live[0] && fmaf(x01[1], af, bf) > 0.f) ? y[q][1] : 0.f;
                    const float g2 = (live[1] && fmaf(x23[0], af, bf) > 0.f) ? y[q][2] : 0.f;
                    const float g3 = (live[1] && fmaf(x23[1], af, bf) > 0.f) ? y[q][3] : 0.f;
                    sv[2 * (q + 8 * h)] = (g0 + g1) + (g2 + g3);
                    sv[2 * (q + 8 * h) + 1] = fmaf(g0, (x01[0] - mf) * is, g1 * ((x01[1] - mf) * is)) +
                                              fmaf(g2, (x23[0] - mf) * is, g3 * ((x23[1] - mf) * is));
                }
            } else if (a.stats && a.stats_mode == 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float y0 = live[0] ? y[q][0] : 0.f, y1 = live[0] ? y[q][1] : 0.f;
                    const float y2 = live[1] ? y[q][2] : 0.f, y3 = live[1] ? y[q][3] : 0.f;
                    sv[2 * (q + 8 * h)] = (y0 + y1) + (y2 + y3);
                    sv[2 * (q + 8 * h) + 1] = fmaf(y0, y0, y1 * y1) + fmaf(y2, y2, y3 * y3);
                }
            }
#undef NEF_ROW
        }
        if (slot_out) {      // halving butterfly over the 32 lanes that share `hi` (conv_wino4_kernel): lane lo ends with value lo
#pragma unroll
            for (int step = 0; step < 5; ++step) {
                const int off = 16 >> step;
                const bool up = (lo & off) != 0;
#pragma unroll
                for (int k = 0; k < off; ++k) {
                    const float send = up ? sv[k] : sv[k + off];
                    const float keep = up ? sv[k + off] : sv[k];
                    sv[k] = keep + __shfl_xor(send, off, 64);
                }
            }
            const int r = lo >> 1;
            const int ch = g * Cog + cobase + (r & 3) + 8 * (r >> 2);
            const int64_t nslot = (int64_t)tps * 2;
            const int64_t slot = (int64_t)b0 * nslot + (int64_t)(t0 / NTO) * 2 + wn;
            if (inb) slot_out[((int64_t)ch * a.B * nslot + slot) * 2 + (lo & 1)] = sv[0];
        }
    }
#if NEF_H2_EPI_ALWAYS
    __syncthreads();
    return over_ret;
#else
    return -1.f;
#endif
#undef a
    };      // tile_pass
    kargs_t* ap_ = (kargs_t*)__builtin_amdgcn_kernarg_segment_ptr();      // (a_ is the segment's first member)
    (void)a_;
    const float over_ = tile_pass(ap_, (int)blockIdx.x, (int)threadIdx.x, tps_, n_tiles_, m_tiles_, 0.f, false);
#if NEF_H2_RESCUE
    if (over_ > 0.f) {
        int e_;
        (void)frexpf(over_, &e_);
        int bid_ = (int)blockIdx.x, tid_ = (int)threadIdx.x, tps2_ = tps_, nt2_ = n_tiles_, mt2_ = m_tiles_;
        asm volatile("" : "+s"(ap_), "+s"(bid_), "+s"(tps2_), "+s"(nt2_), "+s"(mt2_));
        asm volatile("" : "+v"(tid_));
        (void)tile_pass(ap_, bid_, tid_, tps2_, nt2_, mt2_, ldexpf(1.f, 9 - e_ < 100 ? 9 - e_ : 100), true);
    }
#endif
}

template <int K, int PRO, int TM, bool PACK = false>
int launch_h2(const nef_conv_args& a, hipStream_t st) {
    constexpr int MT = 64 * TM;
    constexpr int XROW = NTO + K - 1;
    constexpr int P4 = (XROW + 3) / 4 + 1;
    constexpr int PLANE = 4 * P4 * 32;
    constexpr size_t lds = (size_t)4 * PLANE + (((PRO & 1) ? 2 * PRO_MAX_CIN : 0) + 8 * MT + 4) * sizeof(float);
    static unsigned long long lds_set = 0;
    if (int e = nef_ensure_dyn_lds(reinterpret_cast<const void*>(&conv_h2_kernel<K, PRO, TM, PACK>), lds, &lds_set)) return e;
    // PACK: `tps` = samples per tile (pitch T + 4; the last sample needs no gap behind it)
    const int tps = PACK ? (NTO + 4) / (a.T + 4) : (a.T + NTO - 1) / NTO;
    const int n_tiles = PACK ? (a.B + tps - 1) / tps : a.B * tps;
    const int m_tiles = a.Cout_g / MT;
    const int64_t blocks = (int64_t)a.G * m_tiles * n_tiles;
    if (blocks <= 0 || blocks > 0x7fffffff) return NEF_E_SHAPE;
    hipLaunchKernelGGL((conv_h2_kernel<K, PRO, TM, PACK>), dim3((unsigned)blocks), dim3(256), lds, st, a, tps, n_tiles, m_tiles);
    return nef_launch_status();
}

}  // namespace

// ---- entry points of this file (hidden: reached through nef_conv_fwd / nef_pack_weights / nef_pack_weight_h2)
// short rows (PACK): several samples per tile; plain launches only (no prologue, channel scale, statistics or BatchNorm-backward sums)
static bool h2_pack_shape(const nef_conv_args* a) {
    if (!(a->T >= 8 && a->T <= 64 && a->T % 4 == 0 && (a->K == 1 || a->K == 3))) return false;
    const int64_t spt = (NTO + 4) / (a->T + 4);
    const int64_t lim = 0x7fffffff / 4;
    return a->pro_mode == 0 && !a->in_scale && !a->res_scale && !a->gate_rowscale && !a->stats && !a->bnb_slots && spt * a->x_bs < lim && spt * a->y_bs < lim &&
           (!a->res || spt * a->res_bs < lim) && (!a->gate || spt * a->gate_bs < lim);
}

__attribute__((visibility("hidden"))) bool nef_h2_ok(const nef_conv_args* a) {
    return (a->K == 1 || a->K == 3 || a->K == 7) && a->Cout_g % 64 == 0 && a->Cin_g % KC == 0 && a->T % 2 == 0 &&
           (a->T >= NTO / 2 || h2_pack_shape(a)) &&
           ((a->pro_mode >= 0 && a->pro_mode <= 4) || a->pro_mode == 8 || a->pro_mode == 9) && (a->K == 3 || a->pro_mode == 0) &&
           !(a->pro_mode && a->in_scale) &&
           (!a->res_scale || (a->res && !(a->pro_mode & 8))) &&
           ((!a->gate_rowscale && a->stats_mode == 0) || (a->gate && !(a->pro_mode & 8) && (a->stats_mode == 0 || (a->stats_mode == 1 && a->stats && !a->bnb_slots)))) &&
           (!(a->pro_mode & 8) || (a->Cout_g % 128 == 0 && a->T >= NTO / 2 && !a->res && !a->gate && !a->mask && !a->relu && a->drop_p <= 0.f &&
                                   !a->bnb_slots && (int64_t)a->Cout_g * a->T * 4 < 0x7fffffff)) &&
           (a->pro_mode != 4 || (a->T >= NTO / 2 && (int64_t)a->Cin_g * a->T * 4 < 0x7fffffff)) &&
           (!(a->pro_mode & 1) || a->Cin_g <= PRO_MAX_CIN);
}

// Kernel-form options of the process (nef_set_option / nef_get_option; under NEF_DIAG=1 the environment gives the initial values:
// NEF_H2P, NEF_H2P_WGS).  The producer / consumer form they select (tools/experiments/conv_h2p.hip: bit-identical, measured slower,
// DESIGN.md 3.0a) is only linked into builds made with `csrc/build.py --with-experiments`: the two hooks below are weak, and a
// library without them keeps every launch on conv_h2_kernel whatever the option says.
static int g_opt[4] = {0, -1, -1, 0};
static void opt_init() {
    if (__atomic_load_n(&g_opt[0], __ATOMIC_ACQUIRE)) return;
    const char* e1 = nef_diag_env("NEF_H2P");
    const char* e2 = nef_diag_env("NEF_H2P_WGS");
    int v1 = e1 ? atoi(e1) : 0, v2 = e2 ? atoi(e2) : 1;
    int neg = -1;
    __atomic_compare_exchange_n(&g_opt[NEF_OPT_H2_FORM], &neg, v1, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
    neg = -1;
    __atomic_compare_exchange_n(&g_opt[NEF_OPT_H2P_WGS], &neg, v2, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
    __atomic_store_n(&g_opt[0], 1, __ATOMIC_RELEASE);
}
__attribute__((visibility("hidden"))) int nef_opt_get(int key) {
    opt_init();
    return __atomic_load_n(&g_opt[key], __ATOMIC_RELAXED);
}
extern "C" {
int nef_set_option(int key, int value) {
    if (key != NEF_OPT_H2_FORM && key != NEF_OPT_H2P_WGS) return NEF_E_SHAPE;
    opt_init();
    return __atomic_exchange_n(&g_opt[key], value, __ATOMIC_RELAXED);
}
int nef_get_option(int key) {
    if (key != NEF_OPT_H2_FORM && key != NEF_OPT_H2P_WGS) return NEF_E_SHAPE;
    return nef_opt_get(key);
}
}
__attribute__((weak, visibility("hidden"))) bool nef_h2p_ok(const nef_conv_args* a);
__attribute__((weak, visibility("hidden"))) int nef_h2p_launch(const nef_conv_args* a, hipStream_t st);

__attribute__((visibility("hidden"))) int nef_h2_launch(const nef_conv_args* a, hipStream_t st) {
    if (!nef_h2_ok(a)) return NEF_E_SHAPE;
    if ((a->pro_mode & 1) && !(a->pro_a && a->pro_b && a->pro_Bp > 0)) return NEF_E_NULL;
    if (&nef_h2p_launch && &nef_h2p_ok && nef_opt_get(NEF_OPT_H2_FORM) && a->pro_mode <= 3 && !a->res_scale && !a->gate_rowscale && a->stats_mode == 0 && nef_h2p_ok(a))
        return nef_h2p_launch(a, st);
    static const bool force_tm1 = nef_diag_env("NEF_H2_TM1") && atoi(nef_diag_env("NEF_H2_TM1")) == 1;      // A/B: 64-channel tile everywhere
    const bool wide = a->Cout_g % 128 == 0 && !force_tm1;
    // the x2-upsampling prologue keeps two source samples per staged position in registers: next to the 128 accumulator
    // registers of the 128-channel tile that spills (250..330 bytes per lane; 34.1 vs 31.1 ms/step in round 4), so those launches
    // always take the 64-channel tile (the 128-channel instantiations <3, 2, 2> / <3, 3, 2> are no longer built)
    if (a->T < NTO / 2)      // short rows: the 64-channel tile (the 128-channel form spills 185 registers with the per-lane sample offsets)
        return a->K == 1 ? launch_h2<1, 0, 1, true>(*a, st) : launch_h2<3, 0, 1, true>(*a, st);
    if (a->K == 7) return wide ? launch_h2<7, 0, 2>(*a, st) : launch_h2<7, 0, 1>(*a, st);
    if (a->K == 1) return wide ? launch_h2<1, 0, 2>(*a, st) : launch_h2<1, 0, 1>(*a, st);
    switch (a->pro_mode) {
        case 0: return wide ? launch_h2<3, 0, 2>(*a, st) : launch_h2<3, 0, 1>(*a, st);
        case 1: return wide ? launch_h2<3, 1, 2>(*a, st) : launch_h2<3, 1, 1>(*a, st);
        case 2: return launch_h2<3, 2, 1>(*a, st);
        case 4: return wide ? launch_h2<3, 4, 2>(*a, st) : launch_h2<3, 4, 1>(*a, st);
        case 8: return launch_h2<3, 8, 2>(*a, st);
        case 9: return launch_h2<3, 9, 2>(*a, st);
        default: return launch_h2<3, 3, 1>(*a, st);
    }
}

__attribute__((visibility("hidden"))) int nef_h2_pack(const nef_pack_desc* descs, int n, hipStream_t st) {
    for (int i0 = 0; i0 < n; i0 += H2_PACK_MAX) {
        H2PackTable tab;
        const int m = n - i0 < H2_PACK_MAX ? n - i0 : H2_PACK_MAX;
        int rows_max = 1;
        for (int i = 0; i < m; ++i) {
            const nef_pack_desc& d = descs[i0 + i];
            if (!d.w || !d.wp) return NEF_E_NULL;
            const int co_n = d.transpose_flip ? d.Cig : d.Cog, ci_n = d.transpose_flip ? d.Cog : d.Cig;
            if (d.G <= 0 || co_n % 32 != 0 || ci_n % 16 != 0 || (d.K != 1 && d.K != 3 && d.K != 7)) return NEF_E_SHAPE;
            if (d.src_mode != 0 && !(d.src_mode == 1 && d.K == 3 && d.Cog % 2 == 0 && d.src_Cr >= 0 &&
                                     (d.src_Cr == 0 || (d.src_Cr % 64 == 0 && d.Cog == 2 * d.src_Cr)))) return NEF_E_SHAPE;
            tab.d[i] = H2PackDesc{d.w, reinterpret_cast<_Float16*>(d.wp), d.G, d.Cog, d.Cig, d.K, d.transpose_flip, d.src_mode, d.src_Cr};
            if (d.G * co_n > rows_max) rows_max = d.G * co_n;
        }
        hipLaunchKernelGGL(pack_h2_kernel, dim3((unsigned)((rows_max + 3) / 4), (unsigned)m), dim3(256), 0, st, tab);
    }
    return nef_launch_status();
}
