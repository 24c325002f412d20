// conv_h2w.hip -- the weight gradient of the grouped 1-D convolutions (K = 3, K = 7) on two-term fp16 splits (fp32-class: 22..23 bits, block-scaled) of BOTH fp32
// operands (see conv_h2.hip for the arithmetic: x = xh + xl, gy = gh + gl, three v_mfma_f32_32x32x16_f16 per product --
// gh*xh + gh*xl + gl*xh -- fp32 accumulation):
//
//     gw[g][co][ci][k] = sum_{b, t} gy[b][g][co][t] * X[b][g][ci][t + k - PAD],      X = prologue(x) * in_scale, zero padded
//
// The reduction index t is the k dimension of the matrix instruction (16 columns per instruction): a lane of the A fragment
// holds gy[co][t0 + 8 hi .. + 7], a lane of the B fragment X[ci][t0 + 8 hi + k - PAD .. + 7] -- for every tap the SAME row
// shifted by one column.  The tiles are staged in their natural [channel][t] order as fp16 hi / lo planes.  The (sample,
// 64-column tile) sequence is split S ways (nef_h2w_splits: as many workgroups as are resident at once, never one more); every
// workgroup streams its share through two LDS stages and leaves its partial sums in ws[split][g][k][co][ci], which
// conv_bwd_weight_reduce (conv_mfma.hip) adds up in a fixed order.  The workgroups that share a (group, split) -- the co x ci
// tiles of the layer -- run on ONE XCD, so gy and X come from HBM once per split.
//
// Two kernels:
//   conv_h2w2_kernel (below, "second form"; Cout_g % 128 == 0): 128 x 64 channels per workgroup, producer waves stage and split
//           the tiles, consumer waves only issue matrix instructions -- see the comment in front of it;
//   conv_h2w_kernel ("first form"; the 64-output-channel layers): 64 x 64 channels, 256 threads, every wave stages, splits and
//           multiplies in turn, two workgroups per CU.  A lane reads the 16-column window X[ci][t0 + 8 hi - 4 .. + 11] once per 16
//           columns and cuts each tap's fragment out of it in registers (even shifts: register renames, odd shifts: four
//           v_alignbit_b32).  (NEF_H2W_MCO=2 builds its 128 x 64 variant with the taps / chunks divided between two wave groups:
//           measured a wash, kept for A/B.)
//
// Both operands are scaled by exact powers of two derived from the magnitudes their call site measured before (x_amax,
// gy_amax; ops.py keeps them per site as for the forward launches); the product of the two scales is divided out of the
// partial sums.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "nefnet_hip.h"
#include "nef_common.h"

#ifndef NEF_H2_CLAMP
#define NEF_H2_CLAMP 0      // 1: clamp operands at fp16's range before the split (round 4; the range rescue makes it unnecessary)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int TT = 64;                 // reduction columns per staged tile
constexpr int GP = TT * 2 + 8;         // gy row pitch in bytes (136 = 8 x 17)
constexpr int XW = TT + 8;             // staged X columns per row: t0 - 4 .. t0 + TT + 3
constexpr int XP = XW * 2 + 8;         // X row pitch in bytes (152 = 8 x 19)
constexpr int X_PLANE = 64 * XP;
constexpr int gy_plane(int MCO) { return 64 * MCO * GP; }
constexpr int buf_bytes(int MCO) { return 2 * gy_plane(MCO) + 2 * X_PLANE; }      // one stage: gy hi | gy lo | X hi | X lo

struct H2WArgs {
    const float* x;
    const float* gy;
    const float* in_scale;
    const float* pro_a;
    const float* pro_b;
    float* ws;
    const float* x_amax;
    const float* gy_amax;
    float* x_amax_next;
    float* gy_amax_next;
    int* clamped;
    int64_t x_bs, x_gs, gy_bs, gy_gs, sc_bs, sc_gs;
    int64_t x_end, gy_end;            // bytes from x / gy to the end of the last row the launch may touch
    int B, T, G, Cig, Cog, pro_Bp, S, tps, n_tiles, m_tiles, c_tiles, teams;
    float x_scale, gy_scale;
    int xclamp;      // pro_mode bit 2 (polyphase weight gradient): the window's columns -1 and T hold x[0] and x[T-1] instead of zeros
};


// (x0, x1) * s -> fp16 pair h, residual pair l; |x| is clamped at lim = 65000 / s first
__device__ __forceinline__ void split_pair_s(float x0, float x1, float s, float lim, unsigned& h, unsigned& l) {
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
// one value: the low halves of h and l (the high halves are not defined)
__device__ __forceinline__ void split_one_s(float x0, float s, float lim, unsigned& h, unsigned& l) {
#if NEF_H2_CLAMP
    x0 = __builtin_amdgcn_fmed3f(x0, -lim, lim);
#else
    (void)lim;
#endif
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(h) : "v"(x0), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(s), "v"(h));
}

// scale that puts a measured magnitude at [2^8, 2^9)
__device__ __forceinline__ float scale_for(float m) {
    int e;
    (void)frexpf(m, &e);
    return ldexpf(1.f, 9 - e);
}

__device__ __forceinline__ float scale_from(const float* amax, float fallback) {
    float s = fallback != 0.f ? fallback : 1.f;
    if (amax) {
        const float m = amax[0];
        if (m > 0.f && m < 3e38f) {
            int e;
            (void)frexpf(m, &e);
            s = ldexpf(1.f, 9 - e);       // largest magnitude -> [2^8, 2^9), as conv_h2_kernel
        }
    }
    return s;
}

// 16 reduction columns of one tile: taps [K0, K1) of this wave's MCO x (K1 - K0) accumulator tiles
template <int K, int K0, int K1, int MCO, int KA>
__device__ __forceinline__ void h2w_chunk(const unsigned char* ga, const unsigned char* xa, int c, f32x16 (&acc)[MCO][KA]) {
    constexpr int PAD = (K - 1) / 2;
    constexpr int GY_PLANE = gy_plane(MCO);
    // A: gy[co][16 c + 8 hi .. + 7] of the wave's MCO row tiles, both planes
    h16x8 fah[MCO], fal[MCO];
#pragma unroll
    for (int i = 0; i < MCO; ++i) {
        unsigned ah[4], al[4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const u32x2 vh = *reinterpret_cast<const u32x2*>(ga + i * 32 * GP + c * 32 + 8 * j);
            const u32x2 vl = *reinterpret_cast<const u32x2*>(ga + i * 32 * GP + GY_PLANE + c * 32 + 8 * j);
            ah[2 * j] = vh[0], ah[2 * j + 1] = vh[1];
            al[2 * j] = vl[0], al[2 * j + 1] = vl[1];
        }
        fah[i] = __builtin_bit_cast(h16x8, u32x4{ah[0], ah[1], ah[2], ah[3]});
        fal[i] = __builtin_bit_cast(h16x8, u32x4{al[0], al[1], al[2], al[3]});
    }
    // B window: X columns 16 c + 8 hi .. + 15 of the staged row (= t0 + 16 c + 8 hi - 4 .. + 11)
    unsigned wh[8], wl[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u32x2 vh = *reinterpret_cast<const u32x2*>(xa + c * 32 + 8 * j);
        const u32x2 vl = *reinterpret_cast<const u32x2*>(xa + X_PLANE + c * 32 + 8 * j);
        wh[2 * j] = vh[0], wh[2 * j + 1] = vh[1];
        wl[2 * j] = vl[0], wl[2 * j + 1] = vl[1];
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int k = K0; k < K1; ++k) {
        const int e0 = 4 + k - PAD;            // first window element of this tap's fragment (compile time after unrolling)
        unsigned bh[4], bl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if ((e0 & 1) == 0) {
                bh[j] = wh[e0 / 2 + j];
                bl[j] = wl[e0 / 2 + j];
            } else {
                bh[j] = __builtin_amdgcn_alignbit(wh[(e0 + 1) / 2 + j], wh[(e0 - 1) / 2 + j], 16);
                bl[j] = __builtin_amdgcn_alignbit(wl[(e0 + 1) / 2 + j], wl[(e0 - 1) / 2 + j], 16);
            }
        }
        const h16x8 fbh = __builtin_bit_cast(h16x8, u32x4{bh[0], bh[1], bh[2], bh[3]});
        const h16x8 fbl = __builtin_bit_cast(h16x8, u32x4{bl[0], bl[1], bl[2], bl[3]});
#pragma unroll
        for (int i = 0; i < MCO; ++i) acc[i][k - K0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[i], fbh, acc[i][k - K0], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MCO; ++i) acc[i][k - K0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[i], fbl, acc[i][k - K0], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MCO; ++i) acc[i][k - K0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[i], fbh, acc[i][k - K0], 0, 0, 0);
    }
}

// SPLIT: 0 = 256 threads, every wave all taps and chunks; 1 = 512 threads, the two wave groups split the taps (K = 7: 4 + 3);
// 2 = 512 threads, the two wave groups take alternate 16-column chunks and leave two partial sums (split index 2 sp + group)
template <int K, int PRO, int MCO, int SPLIT>
__global__ __launch_bounds__(SPLIT ? 512 : 256, SPLIT ? 1 : 2) void conv_h2w_kernel(H2WArgs a) {
    constexpr bool UP = (PRO & 2) != 0, AFF = (PRO & 1) != 0;
    constexpr int NT = SPLIT ? 512 : 256;
    constexpr int ROWS_G = 64 * MCO;
    constexpr int GY_PLANE = gy_plane(MCO), BUF = buf_bytes(MCO);
    constexpr int KA = SPLIT == 1 ? (K + 1) / 2 : K;      // accumulator tiles per row tile and wave
    constexpr int NGQ = ROWS_G * (TT / 2) / NT;           // gy column pairs per thread and tile
    constexpr int NXQ = (64 * (XW / 2) + NT - 1) / NT;    // X column pairs per thread and tile
    static_assert(SPLIT == 0 || MCO == 2, "the 512-thread forms are the 128-channel tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];

    // team = (group, split); its members = the co x ci tiles.  Block ids go round-robin over the 8 XCDs: the members of a team
    // take ids that are congruent mod 8
    const int members = a.m_tiles * a.c_tiles;
    const int xcd = blockIdx.x & 7, q_ = blockIdx.x >> 3;
    const int member = q_ % members;
    const int team = (q_ / members) * 8 + xcd;
    if (team >= a.teams) return;
    const int g = team / a.S, sp = team % a.S;
    const int mt = member / a.c_tiles, ct = member % a.c_tiles;
    const int T = a.T, Tin = UP ? (T >> 1) : T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wt = wave_u >> 2;                        // wave group (0 with 256 threads)
    const int wm = (wave_u >> 1) & 1, wn = wave_u & 1;

    const int64_t per = ((int64_t)a.n_tiles + a.S - 1) / a.S;
    const int n_lo = (int)(per * sp);
    int n_hi = (int)(per * (sp + 1));
    if (n_hi > a.n_tiles) n_hi = a.n_tiles;

    // Range rescue (round 5, as conv_h2_kernel): the workgroup's whole share is redone with scales from its OWN operands if an element
    // left fp16's range under the launch's scales; the pass is a lambda inlined twice (a backward branch demotes the arrays to
    // scratch).  Returns true when the partial sums were written.
    float* const Wl = reinterpret_cast<float*>(smem_w);      // [2][8] wave magnitudes, written behind the last tile (the stages are free then)
    float rx_ = 0.f, rg_ = 0.f;                              // the share's magnitudes, left by a pass that did not fit
    auto share_pass = [&](const float sx, const float sg, const bool last) __attribute__((always_inline)) -> bool {
    const float limx = 65000.f / sx, limg = 65000.f / sg;
    float amax_x = 0.f, amax_g = 0.f;

    f32x16 acc[MCO][KA];
#pragma unroll
    for (int i = 0; i < MCO; ++i)
#pragma unroll
        for (int k = 0; k < KA; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;

    // ---- staging registers: this thread's column pairs of the tile in flight
    f32x2 gq[NGQ];
    f32x2 xq[NXQ][UP ? 2 : 1];       // UP: the two half-resolution source pairs (see stage_x)
    // pair p of the gy tile: row p / 32, columns 2 (p % 32) ..; pair p of the X tile: row p / 36, columns 2 (p % 36) .. (t0 - 4 + ..)
#define NEF_W_ISSUE(N)                                                                                               \
    {                                                                                                               \
        const int b_ = (N) / a.tps, t0_ = ((N) % a.tps) * TT;                                                       \
        const bool on_ = (N) < n_hi;                                                                                \
        const __amdgpu_buffer_rsrc_t grs = nef_rsrc_n(a.gy + (int64_t)b_ * a.gy_bs + (int64_t)g * a.gy_gs + (int64_t)(mt * ROWS_G) * T, \
                                                      on_ ? 0x7FFFFFFCu : 0u);                                      \
        _Pragma("unroll") for (int q = 0; q < NGQ; ++q) {                                                           \
            const int p = (int)threadIdx.x + NT * q;                                                                \
            const int row = p >> 5, t = t0_ + 2 * (p & 31);                                                         \
            gq[q] = nef_buf_f32x2(grs, t < T ? (unsigned)((row * T + t) * 4) : NEF_OOB, 0);                         \
        }                                                                                                           \
        const __amdgpu_buffer_rsrc_t xrs = nef_rsrc_n(a.x + (int64_t)b_ * a.x_bs + (int64_t)g * a.x_gs + (int64_t)(ct * 64) * Tin, \
                                                      on_ ? 0x7FFFFFFCu : 0u);                                      \
        _Pragma("unroll") for (int q = 0; q < NXQ; ++q) {                                                           \
            const int p = (int)threadIdx.x + NT * q;                                                                \
            const int row = p / (XW / 2), t = t0_ - 4 + 2 * (p % (XW / 2));                                         \
            const bool ok = p < 64 * (XW / 2) && t >= 0 && t < T;                                                   \
            if constexpr (UP) {                                                                                     \
                /* outputs t, t + 1 (t even): sources t/2 - 1, t/2 and t/2, t/2 + 1, clamped as nn.Upsample clamps */ \
                const int h_ = t >> 1;                                                                              \
                const int im = h_ > 0 ? h_ - 1 : 0, ip = h_ + 1 < Tin ? h_ + 1 : Tin - 1;                           \
                const float v0 = nef_buf_f32(xrs, ok ? (unsigned)((row * Tin + im) * 4) : NEF_OOB, 0);              \
                const float v1 = nef_buf_f32(xrs, ok ? (unsigned)((row * Tin + h_) * 4) : NEF_OOB, 0);              \
                const float v2 = nef_buf_f32(xrs, ok ? (unsigned)((row * Tin + ip) * 4) : NEF_OOB, 0);              \
                xq[q][0] = f32x2{v0, v1};                                                                           \
                xq[q][UP ? 1 : 0] = f32x2{v1, v2};                                                                  \
            } else {                                                                                                \
                xq[q][0] = nef_buf_f32x2(xrs, ok ? (unsigned)((row * T + t) * 4) : NEF_OOB, 0);                     \
            }                                                                                                       \
        }                                                                                                           \
    }
#define NEF_W_STORE(N, BUFP)                                                                                         \
    {                                                                                                               \
        const int b_ = (N) / a.tps, t0_ = ((N) % a.tps) * TT;                                                       \
        _Pragma("unroll") for (int q = 0; q < NGQ; ++q) {                                                           \
            const int p = (int)threadIdx.x + NT * q;                                                                \
            const int row = p >> 5, c2 = p & 31;                                                                    \
            amax_g = fmaxf(amax_g, fmaxf(fabsf(gq[q][0]), fabsf(gq[q][1])));                                        \
            unsigned h_, l_;                                                                                        \
            split_pair_s(gq[q][0], gq[q][1], sg, limg, h_, l_);                                                           \
            unsigned char* p_ = (BUFP) + row * GP + c2 * 4;                                                         \
            *reinterpret_cast<unsigned*>(p_) = h_;                                                                  \
            *reinterpret_cast<unsigned*>(p_ + GY_PLANE) = l_;                                                       \
        }                                                                                                           \
        _Pragma("unroll") for (int q = 0; q < NXQ; ++q) {                                                           \
            const int p = (int)threadIdx.x + NT * q;                                                                \
            const int row = p / (XW / 2), c2 = p % (XW / 2), t = t0_ - 4 + 2 * c2;                                  \
            if (p < 64 * (XW / 2)) {                                                                                \
                const int ch = g * a.Cig + ct * 64 + row;                                                           \
                float v0, v1;                                                                                       \
                if constexpr (UP) {                                                                                 \
                    float s0 = xq[q][0][0], s1 = xq[q][0][1], s2 = xq[q][UP ? 1 : 0][1];                            \
                    if constexpr (AFF) {                                                                            \
                        const int pr = (b_ / a.pro_Bp) * a.G * a.Cig + ch;                                          \
                        const float pa = a.pro_a[pr], pb = a.pro_b[pr];                                             \
                        s0 = fmaxf(fmaf(s0, pa, pb), 0.f), s1 = fmaxf(fmaf(s1, pa, pb), 0.f), s2 = fmaxf(fmaf(s2, pa, pb), 0.f); \
                    }                                                                                               \
                    /* t even: src = t/2 - 0.25 -> 0.25 x[t/2-1] + 0.75 x[t/2] (t = 0: clamped to x[0]); t + 1: 0.75 x[t/2] + 0.25 x[t/2+1] */ \
                    v0 = t > 0 ? (1.f - 0.75f) * s0 + 0.75f * s1 : s1;                                              \
                    v1 = (1.f - 0.25f) * s1 + 0.25f * s2;                                                           \
                } else {                                                                                            \
                    v0 = xq[q][0][0], v1 = xq[q][0][1];                                                             \
                    if constexpr (AFF) {                                                                            \
                        const int pr = (b_ / a.pro_Bp) * a.G * a.Cig + ch;                                          \
                        const float pa = a.pro_a[pr], pb = a.pro_b[pr];                                             \
                        v0 = fmaxf(fmaf(v0, pa, pb), 0.f), v1 = fmaxf(fmaf(v1, pa, pb), 0.f);                       \
                    }                                                                                               \
                }                                                                                                   \
                if (a.in_scale) {                                                                                   \
                    const float sc = a.in_scale[(int64_t)b_ * a.sc_bs + (int64_t)g * a.sc_gs + ct * 64 + row];      \
                    v0 *= sc, v1 *= sc;                                                                             \
                }                                                                                                   \
                if (t < 0 || t >= T) v0 = v1 = 0.f;      /* zero padding comes after the prologue */                \
                amax_x = fmaxf(amax_x, fmaxf(fabsf(v0), fabsf(v1)));                                                \
                unsigned h_, l_;                                                                                    \
                split_pair_s(v0, v1, sx, limx, h_, l_);                                                                   \
                unsigned char* p_ = (BUFP) + 2 * GY_PLANE + row * XP + c2 * 4;                                      \
                *reinterpret_cast<unsigned*>(p_) = h_;                                                              \
                *reinterpret_cast<unsigned*>(p_ + X_PLANE) = l_;                                                    \
            }                                                                                                       \
        }                                                                                                           \
    }

    if (n_lo < n_hi) {
        NEF_W_ISSUE(n_lo)
        NEF_W_STORE(n_lo, smem_w)
    }
    __syncthreads();
    for (int n = n_lo; n < n_hi; ++n) {
        const unsigned char* const bufp = smem_w + ((n - n_lo) & 1) * BUF;
        NEF_W_ISSUE(n + 1)
        const unsigned char* const ga = bufp + (wm * MCO * 32 + lo) * GP + hi * 16;
        const unsigned char* const xa = bufp + 2 * GY_PLANE + (wn * 32 + lo) * XP + hi * 16;
        if constexpr (SPLIT == 1) {
            // (chunk loops NOT unrolled: with both tap groups' code in one kernel the unrolled form spilled 90 registers)
            if (wt == 0) {
#pragma unroll 1
                for (int c = 0; c < TT / 16; ++c) h2w_chunk<K, 0, KA, MCO, KA>(ga, xa, c, acc);
            } else {
#pragma unroll 1
                for (int c = 0; c < TT / 16; ++c) h2w_chunk<K, KA, K, MCO, KA>(ga, xa, c, acc);
            }
        } else if constexpr (SPLIT == 2) {
#pragma unroll
            for (int cc = 0; cc < TT / 32; ++cc) h2w_chunk<K, 0, K, MCO, KA>(ga, xa, 2 * cc + wt, acc);
        } else {
#pragma unroll
            for (int c = 0; c < TT / 16; ++c) h2w_chunk<K, 0, K, MCO, KA>(ga, xa, c, acc);
        }
        if (n + 1 < n_hi) NEF_W_STORE(n + 1, smem_w + ((n + 1 - n_lo) & 1) * BUF)
        __syncthreads();
    }
#undef NEF_W_ISSUE
#undef NEF_W_STORE

    // this share's operand magnitudes: for the rescue, and for the call site's next launch
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        amax_x = fmaxf(amax_x, __shfl_xor(amax_x, o, 64));
        amax_g = fmaxf(amax_g, __shfl_xor(amax_g, o, 64));
    }
    if (lane == 0) Wl[wave_u] = amax_x, Wl[8 + wave_u] = amax_g;
    __syncthreads();
    {
        float wx = 0.f, wg_ = 0.f;
#pragma unroll
        for (int i = 0; i < NT / 64; ++i) wx = fmaxf(wx, Wl[i]), wg_ = fmaxf(wg_, Wl[8 + i]);
        __syncthreads();      // (the next pass stages over Wl)
        if (!last && !(wx * sx < 65000.f && wg_ * sg < 65000.f) && wx < 3e38f && wg_ < 3e38f) {
            rx_ = wx, rg_ = wg_;
            return false;
        }
    }
    if (a.clamped && lane == 0 && !(amax_x * sx < 65000.f && amax_g * sg < 65000.f)) atomicAdd(a.clamped, 1);      // not finite
    if (a.x_amax_next && lane == 0) {
        unsigned* const px = reinterpret_cast<unsigned*>(a.x_amax_next);
        unsigned* const pg = reinterpret_cast<unsigned*>(a.gy_amax_next);
        const unsigned bx = __builtin_bit_cast(unsigned, amax_x), bg = __builtin_bit_cast(unsigned, amax_g);
        if (amax_x < 3e38f && bx > __atomic_load_n(px, __ATOMIC_RELAXED)) atomicMax(px, bx);
        if (amax_g < 3e38f && bg > __atomic_load_n(pg, __ATOMIC_RELAXED)) atomicMax(pg, bg);
    }

    // partial sums: ws[split][g][k][co][ci]; a lane's column is ci = wn 32 + lo, its rows co = (wm MCO + i) 32 + 4 hi + (r & 3) + 8 (r >> 2)
    const float ds = 1.f / (sx * sg);
    const int ci = ct * 64 + wn * 32 + lo;
    const int spw = SPLIT == 2 ? 2 * sp + wt : sp;
    const int k0 = SPLIT == 1 ? wt * KA : 0;
    float* const wsp = a.ws + ((int64_t)spw * a.G + g) * K * a.Cog * a.Cig;
#pragma unroll
    for (int i = 0; i < MCO; ++i) {
        const int co0 = mt * ROWS_G + (wm * MCO + i) * 32 + 4 * hi;
#pragma unroll
        for (int k = 0; k < KA; ++k) {
            if (k0 + k < K) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    wsp[((int64_t)(k0 + k) * a.Cog + co0 + (r & 3) + 8 * (r >> 2)) * a.Cig + ci] = acc[i][k][r] * ds;
            }
        }
    }
    return true;
    };      // share_pass
    const float sx0 = scale_from(a.x_amax, a.x_scale), sg0 = scale_from(a.gy_amax, a.gy_scale);
    if (!share_pass(sx0, sg0, false))
        (void)share_pass(rx_ * sx0 < 65000.f ? sx0 : scale_for(rx_), rg_ * sg0 < 65000.f ? sg0 : scale_for(rg_), true);
}

template <int K, int PRO, int MCO, int SPLIT>
int launch_h2w(const H2WArgs& a, hipStream_t st) {
    constexpr size_t lds = 2 * buf_bytes(MCO);
    static unsigned long long lds_set = 0;
    if (int e = nef_ensure_dyn_lds(reinterpret_cast<const void*>(&conv_h2w_kernel<K, PRO, MCO, SPLIT>), lds, &lds_set)) return e;
    const int members = a.m_tiles * a.c_tiles;
    const int teams8 = (a.teams + 7) / 8 * 8;
    const int64_t blocks = (int64_t)teams8 * members;
    if (blocks <= 0 || blocks > 0x7fffffff) return NEF_E_SHAPE;
    hipLaunchKernelGGL((conv_h2w_kernel<K, PRO, MCO, SPLIT>), dim3((unsigned)blocks), dim3(SPLIT ? 512 : 256), lds, st, a);
    return nef_launch_status();
}

// tile form for a shape: MCO (row tiles of 64 channels per workgroup) and SPLIT (see the kernel)
struct H2WForm { int mco, split; };
H2WForm h2w_form(int Cog, int K) {
    // Measured in round 4 (ms, 64 x 64 tile -> 128 x 64 tile): decoder 128->128 T=2500 0.93 -> 0.85, first decoder layer 1.21 -> 1.11,
    // but encoder-side K = 3 0.54 -> 0.60 and K = 7 0.87 -> 3.3 (spills): a wash; the 128 x 64 instantiations (MCO = 2) are no longer
    // built -- the layers with 128 output channels take the producer / consumer form below
    (void)Cog, (void)K;
    return {1, 0};
}


// =====================================================================================================================
// Second form (Cout_g % 128 == 0): producer and consumer waves.
//
// Timing-only builds of the first form (NEF_W2_DBG, K = 7 encoder shape, random data) showed three costs that ADD: the matrix
// stream alone 0.50 ms (= the rate this chip sustains for dense fp16 MFMA on random data, 1.3 PFLOP/s), staging (split + LDS
// stores) 0.13 ms, exposed global-load time 0.17 ms -- whatever the tile form, as long as every wave does all three in turn.
// Here the roles are separate waves of one workgroup (128 output x 64 input channels, 64-column tiles, two LDS stages, ONE
// barrier per tile):
//   * consumer waves only read fragments and issue matrix instructions.  X is staged TWICE -- in its natural order and shifted
//     by one column (the "odd copy": word j = columns 2j + 1, 2j + 2) -- so a lane reads its 16-column window of each copy with
//     two aligned ds_read_b128 per plane and every tap's fragment is four consecutive dwords of one window: no alignbit; row
//     pitches are 16 x odd bytes (144), conflict-free for the 16-lane groups of ds_read_b128.  The three products of a tap
//     are issued K (x row tiles) matrix instructions apart;
//   * producer waves stage QUADS of columns (one dwordx4 load, + the column behind it for the odd copy) and split them with
//     the scale folded into the conversion: hi = v_fma_mixlo/hi_f16(x, s, 0), lo = v_fma_mixlo/hi_f16(x, s, -hi) -- 4
//     instructions per pair instead of 6, exact as before (x s is exact, x s - hi is exact in fp32).  All index arithmetic of
//     a tile is scalar (per-lane byte offsets are fixed for the launch, the tile's origin goes into the buffer descriptor);
//     tiles that touch a row end take a path with per-column checks (EDGE), the others have no selects at all.
// Measured (tools/_h2w_check.py, ms, first form -> this one): K = 7 encoder 0.76 -> 0.68, w_conv 0.46 -> 0.39, decoder
// 128 -> 128 T = 2500 0.89 -> 0.74, first decoder layer 1.16 -> 0.95.  What remains is the producers' load rate: alone they
// stream 3.0 - 3.6 TB/s (7 - 8 B/clk/CU of 256-byte row pieces that straddle cache lines); the 64-channel layers are at that
// rate with the first form already (64 -> 64, T = 5000: 1.97 GB in 0.50 ms) and stay on it.
// =====================================================================================================================
#ifndef NEF_W2_DBG
#define NEF_W2_DBG 0      // timing-only builds: 1 = no split / LDS stores, 2 = no global loads, 4 = no matrix work
#endif
#ifndef NEF_H2W_64_DEFAULT
#define NEF_H2W_64_DEFAULT 12     // the x2-upsampling prologues (modes 2, 3): 1.12 -> 0.92 ms for the 128 -> 64 decoder layer; plain and affine-only 64-channel layers are at the load rate with either form
#endif
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned extent_from(int64_t end_bytes, int64_t off_elems, bool on) {
    int64_t r = end_bytes - off_elems * 4;
    if (r < 0) r = 0;
    if (r > 0x7FFFFFFC) r = 0x7FFFFFFC;
    return on ? (unsigned)r : 0u;
}

// 16 reduction columns: the wave's K accumulator tiles
template <int K, int WR, int GP, int XP, int G_PLANE, int X_PLANE>
__device__ __forceinline__ void h2w2_chunk(const unsigned char* ga, const unsigned char* xa, int c, f32x16 (&acc)[WR][K]) {
    constexpr int PAD = (K - 1) / 2;
    h16x8 fah[WR], fal[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        fah[i] = __builtin_bit_cast(h16x8, *reinterpret_cast<const u32x4*>(ga + i * 32 * GP + c * 32));
        fal[i] = __builtin_bit_cast(h16x8, *reinterpret_cast<const u32x4*>(ga + i * 32 * GP + G_PLANE + c * 32));
    }
    // B: the lane's 16-column window of both copies (two aligned ds_read_b128 per plane); the fragment of a tap is four
    // consecutive dwords of one of them, starting at dword e0 / 2 (natural copy, e0 even) or (e0 - 1) / 2 (odd copy)
    unsigned wn_h[8], wn_l[8], wo_h[8], wo_l[8];
    constexpr bool NAT = K >= 1, ODD = K >= 3;      // K = 1 only reads the natural copy
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (NAT) {
            const u32x4 vh = *reinterpret_cast<const u32x4*>(xa + c * 32 + 16 * j);
            const u32x4 vl = *reinterpret_cast<const u32x4*>(xa + X_PLANE + c * 32 + 16 * j);
#pragma unroll
            for (int i = 0; i < 4; ++i) wn_h[4 * j + i] = vh[i], wn_l[4 * j + i] = vl[i];
        }
        if (ODD) {
            const u32x4 vh = *reinterpret_cast<const u32x4*>(xa + 2 * X_PLANE + c * 32 + 16 * j);
            const u32x4 vl = *reinterpret_cast<const u32x4*>(xa + 3 * X_PLANE + c * 32 + 16 * j);
#pragma unroll
            for (int i = 0; i < 4; ++i) wo_h[4 * j + i] = vh[i], wo_l[4 * j + i] = vl[i];
        }
    }
    h16x8 fbh[K], fbl[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e0 = 4 + k - PAD;       // first window column of the tap's fragment
        const int d = (e0 & 1) ? (e0 - 1) / 2 : e0 / 2;
        if (e0 & 1) {
            fbh[k] = __builtin_bit_cast(h16x8, u32x4{wo_h[d], wo_h[d + 1], wo_h[d + 2], wo_h[d + 3]});
            fbl[k] = __builtin_bit_cast(h16x8, u32x4{wo_l[d], wo_l[d + 1], wo_l[d + 2], wo_l[d + 3]});
        } else {
            fbh[k] = __builtin_bit_cast(h16x8, u32x4{wn_h[d], wn_h[d + 1], wn_h[d + 2], wn_h[d + 3]});
            fbl[k] = __builtin_bit_cast(h16x8, u32x4{wn_l[d], wn_l[d + 1], wn_l[d + 2], wn_l[d + 3]});
        }
    }
#pragma unroll
    for (int i = 0; i < WR; ++i)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[i], fbh[k], acc[i][k], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < WR; ++i)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[i], fbh[k], acc[i][k], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < WR; ++i)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[i], fbl[k], acc[i][k], 0, 0, 0);
}

// WM x WN tiles of 32 x 32 channels per workgroup, WR row tiles per consumer wave, NP producer waves, DEPTH tiles in flight in
// the producers' registers
template <int K, int PRO, int WM, int WN, int WR, int TTv, int NP, int DEPTH>
__global__ __launch_bounds__(64 * (WM / WR * WN + NP), 1) void conv_h2w2_kernel(H2WArgs a) {
    constexpr bool UP = (PRO & 2) != 0, AFF = (PRO & 1) != 0;
    constexpr int NC = WM / WR * WN;               // consumer waves (matrix work); waves NC .. NC + NP - 1 stage the tiles
    constexpr int NT = 64 * NP, RG = 32 * WM, RX = 32 * WN;
    constexpr int GP = TTv * 2 + 16, XE = TTv + 8, XP = XE * 2;
    constexpr int G_PLANE = RG * GP, X_PLANE = RX * XP, BUF = 2 * G_PLANE + 4 * X_PLANE;
    constexpr int QG = TTv / 4, QX = XE / 4, NGQ = RG * QG / NT, NXQ = (RX * QX + NT - 1) / NT, NCH = TTv / 16;
    constexpr int GROWS = NT / QG;                 // gy rows between a thread's consecutive quads
    static_assert(RG * QG % NT == 0 && GP % 32 == 16 && XP % 32 == 16, "tile form");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];

    const int members = a.m_tiles * a.c_tiles;
    const int xcd = blockIdx.x & 7, q_ = blockIdx.x >> 3;
    const int member = q_ % members;
    const int team = (q_ / members) * 8 + xcd;
    if (team >= a.teams) return;
    const int g = team / a.S, sp = team % a.S;
    const int mt = member / a.c_tiles, ct = member % a.c_tiles;
    const int T = a.T, Tin = UP ? (T >> 1) : T;
    const int lane = (int)threadIdx.x & 63;
    const int lo = lane & 31, hi = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int wm = (wave_u % NC) / WN * WR, wn = wave_u % WN;       // first row tile, column tile of a consumer wave
    const bool producer = wave_u >= NC;
    const int tid = ((int)threadIdx.x - 64 * NC) & (NT - 1);       // staging thread index (producers: 0 .. NT - 1)

    const int64_t per = ((int64_t)a.n_tiles + a.S - 1) / a.S;
    const int n_lo = (int)(per * sp);
    int n_hi = (int)(per * (sp + 1));
    if (n_hi > a.n_tiles) n_hi = a.n_tiles;

    // Range rescue (round 5, as conv_h2w_kernel): the whole share again with scales from its own operands if an element left fp16's
    // range; the pass -- both roles -- is a lambda inlined twice.  The producers know the magnitudes; they publish them through the
    // first words of the (then idle) stages behind the last tile.
    float* const Wl = reinterpret_cast<float*>(smem_w);      // [2][16] wave magnitudes
    float rx_ = 0.f, rg_ = 0.f;
    auto share_pass = [&](const float sx, const float sg, const bool last) __attribute__((always_inline)) -> bool {
    const float limx = 65000.f / sx, limg = 65000.f / sg;
    float amax_x = 0.f, amax_g = 0.f;

    f32x16 acc[WR][K];
#pragma unroll
    for (int i = 0; i < WR; ++i)
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;

    // ---- what a thread stages, fixed for the launch.  gy: quads tid + NT q (row tid / QG + GROWS q, quad tid % QG);  X: quads
    // tid + NT q of the RX x QX quads of the window (columns t0 - 4 + 4 quad ..)
    const int g_row = tid / QG, g_qd = tid % QG;
    const unsigned g_voff = (unsigned)((g_row * T + 4 * g_qd) * 4);
    const unsigned g_lds = (unsigned)(g_row * GP + g_qd * 8);
    unsigned x_voff[NXQ], x_lds[NXQ], x_row[NXQ], x_qd[NXQ];
#pragma unroll
    for (int q = 0; q < NXQ; ++q) {
        const int p = tid + NT * q;
        const int row = p < RX * QX ? p / QX : 0, qd = p < RX * QX ? p % QX : 0;
        x_row[q] = (unsigned)row, x_qd[q] = (unsigned)qd;
        x_voff[q] = p < RX * QX ? (unsigned)((row * Tin + (UP ? 2 : 4) * qd) * 4) : NEF_OOB;
        x_lds[q] = (unsigned)(row * XP + qd * 8);
    }
    const bool has_sc = a.in_scale != nullptr;

    // staging registers of one tile in flight; the producers keep TWO tiles in flight (a load is issued two tiles before its store)
    struct Stage {
        f32x4 gq[NGQ];
        f32x4 xq[NXQ];
        float x5[NXQ], ppa[NXQ], ppb[NXQ], psc[NXQ];
    };
    Stage st0, st1;

    // a tile away from the row ends: the whole window (and, UP, every source column) exists
    auto is_edge = [&](int N) { const int t0_ = (N % a.tps) * TTv; return !(t0_ > 0 && t0_ + TTv + 6 <= T); };

    auto issue = [&](Stage& z, int N, auto edge_c, int part) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_c)::value;
        const int b_ = N / a.tps, t0_ = (N % a.tps) * TTv;
        const bool on_ = N < n_hi;
        // gy: descriptor origin = column t0 of the tile's first row
        const int64_t goff = (int64_t)b_ * a.gy_bs + (int64_t)g * a.gy_gs + (int64_t)(mt * RG) * T + (EDGE ? 0 : t0_);
        const __amdgpu_buffer_rsrc_t grs = nef_rsrc_n(a.gy + goff, extent_from(a.gy_end, goff, on_));
#pragma unroll
        for (int q = 0; q < ((part & 1) ? NGQ : 0); ++q) {
            const unsigned so = (unsigned)(q * GROWS * T * 4);
            if constexpr (!EDGE) {
                z.gq[q] = nef_buf_f32x4(grs, g_voff, so);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int t = t0_ + 4 * g_qd + i;
                    z.gq[q][i] = nef_buf_f32(grs, t < T ? g_voff + (unsigned)((t0_ + i) * 4) : NEF_OOB, so);
                }
            }
        }
        // X: origin = the window's first column (t0 - 4; UP: its first source column (t0 - 4) / 2 - 1); EDGE: the row start
        const int64_t xrow0 = (int64_t)b_ * a.x_bs + (int64_t)g * a.x_gs + (int64_t)(ct * RX) * Tin;
        const int64_t xoff = xrow0 + (EDGE ? 0 : (UP ? (t0_ - 4) / 2 - 1 : t0_ - 4));
        const __amdgpu_buffer_rsrc_t xrs = nef_rsrc_n(a.x + xoff, extent_from(a.x_end, xoff, on_));
#pragma unroll
        for (int q = 0; q < ((part & 2) ? NXQ : 0); ++q) {
            if constexpr (!EDGE) {
                z.xq[q] = nef_buf_f32x4(xrs, x_voff[q], 0);
                if constexpr (!UP) z.x5[q] = nef_buf_f32(xrs, x_voff[q], 16);
            } else {
                const unsigned rowoff = x_voff[q] - (unsigned)((UP ? 2 : 4) * 4) * x_qd[q];      // (row Tin) 4, or NEF_OOB - ..: checked below
                const bool live = x_voff[q] != NEF_OOB;
                const int t = t0_ - 4 + 4 * (int)x_qd[q];
                if constexpr (UP) {
                    // sources t/2 - 1 .. t/2 + 2, clamped as nn.Upsample clamps
                    const int h_ = t >> 1;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        int c_ = h_ - 1 + i;
                        c_ = c_ < 0 ? 0 : (c_ > Tin - 1 ? Tin - 1 : c_);
                        z.xq[q][i] = nef_buf_f32(xrs, live && t + 4 >= 0 && t < T ? rowoff + (unsigned)(c_ * 4) : NEF_OOB, 0);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 5; ++i) {
                        int ti = t + i;
                        if (a.xclamp) ti = ti == -1 ? 0 : (ti == T ? T - 1 : ti);      // nn.Upsample's clamped sources (polyphase form)
                        const float v = nef_buf_f32(xrs, live && ti >= 0 && ti < T ? rowoff + (unsigned)(ti * 4) : NEF_OOB, 0);
                        if (i < 4) z.xq[q][i] = v; else z.x5[q] = v;
                    }
                }
            }
            const int bq = on_ ? b_ : 0;      // (tiles past the share are fetched as zeros; their parameters must stay inside the arrays)
            if constexpr (AFF) {
                const int pr = (bq / a.pro_Bp) * a.G * a.Cig + g * a.Cig + ct * RX;
                z.ppa[q] = a.pro_a[pr + (int)x_row[q]], z.ppb[q] = a.pro_b[pr + (int)x_row[q]];
            }
            if (has_sc) z.psc[q] = a.in_scale[(int64_t)bq * a.sc_bs + (int64_t)g * a.sc_gs + ct * RX + (int)x_row[q]];
        }
    };

    auto store = [&](Stage& z, int N, unsigned char* bufp, auto edge_c, int part) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_c)::value;
        const int t0_ = (N % a.tps) * TTv;
#pragma unroll
        for (int q = 0; q < ((part & 1) ? NGQ : 0); ++q) {
            const f32x4 v = z.gq[q];
            amax_g = fmaxf(amax_g, fmaxf(fabsf(v[0]), fabsf(v[1])));
            amax_g = fmaxf(amax_g, fmaxf(fabsf(v[2]), fabsf(v[3])));
            unsigned h0, l0, h1, l1;
            split_pair_s(v[0], v[1], sg, limg, h0, l0);
            split_pair_s(v[2], v[3], sg, limg, h1, l1);
            unsigned char* p_ = bufp + g_lds + q * GROWS * GP;
            *reinterpret_cast<u32x2*>(p_) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(p_ + G_PLANE) = u32x2{l0, l1};
        }
#pragma unroll
        for (int q = 0; q < ((part & 2) ? NXQ : 0); ++q) {
            if (tid + NT * q < RX * QX) {
                float v[5];
                if constexpr (UP) {
                    float s0 = z.xq[q][0], s1 = z.xq[q][1], s2 = z.xq[q][2], s3 = z.xq[q][3];
                    if constexpr (AFF) {
                        s0 = fmaxf(fmaf(s0, z.ppa[q], z.ppb[q]), 0.f), s1 = fmaxf(fmaf(s1, z.ppa[q], z.ppb[q]), 0.f);
                        s2 = fmaxf(fmaf(s2, z.ppa[q], z.ppb[q]), 0.f), s3 = fmaxf(fmaf(s3, z.ppa[q], z.ppb[q]), 0.f);
                    }
                    // column t (even) = 0.25 x[t/2 - 1] + 0.75 x[t/2] (t = 0: x[0]), column t + 1 = 0.75 x[t/2] + 0.25 x[t/2 + 1]
                    v[0] = (1.f - 0.75f) * s0 + 0.75f * s1;
                    v[1] = (1.f - 0.25f) * s1 + 0.25f * s2;
                    v[2] = (1.f - 0.75f) * s1 + 0.75f * s2;
                    v[3] = (1.f - 0.25f) * s2 + 0.25f * s3;
                    v[4] = (1.f - 0.75f) * s2 + 0.75f * s3;
                    if constexpr (EDGE) {      // column 0 is x[0] itself, in both quads that hold it
                        if (t0_ - 4 + 4 * (int)x_qd[q] == 0) v[0] = s1;
                        if (t0_ + 4 * (int)x_qd[q] == 0) v[4] = s3;
                    }
                } else {
                    v[0] = z.xq[q][0], v[1] = z.xq[q][1], v[2] = z.xq[q][2], v[3] = z.xq[q][3], v[4] = z.x5[q];
                    if constexpr (AFF) {
#pragma unroll
                        for (int i = 0; i < 5; ++i) v[i] = fmaxf(fmaf(v[i], z.ppa[q], z.ppb[q]), 0.f);
                    }
                }
                if (has_sc) {
#pragma unroll
                    for (int i = 0; i < 5; ++i) v[i] *= z.psc[q];
                }
                if constexpr (EDGE) {      // zero padding comes after the prologue
                    const int t = t0_ - 4 + 4 * (int)x_qd[q];
                    const int lo_ = a.xclamp ? -1 : 0, hi_ = a.xclamp ? T + 1 : T;
#pragma unroll
                    for (int i = 0; i < 5; ++i)
                        if (t + i < lo_ || t + i >= hi_) v[i] = 0.f;
                }
                amax_x = fmaxf(amax_x, fmaxf(fabsf(v[0]), fabsf(v[1])));
                amax_x = fmaxf(amax_x, fmaxf(fabsf(v[2]), fabsf(v[3])));
                unsigned h0, l0, h1, l1, h2, l2;
                split_pair_s(v[0], v[1], sx, limx, h0, l0);
                split_pair_s(v[2], v[3], sx, limx, h1, l1);
                split_one_s(v[4], sx, limx, h2, l2);
                unsigned char* p_ = bufp + 2 * G_PLANE + x_lds[q];
                *reinterpret_cast<u32x2*>(p_) = u32x2{h0, h1};
                *reinterpret_cast<u32x2*>(p_ + X_PLANE) = u32x2{l0, l1};
                *reinterpret_cast<u32x2*>(p_ + 2 * X_PLANE) = u32x2{__builtin_amdgcn_alignbit(h1, h0, 16), __builtin_amdgcn_alignbit(h2, h1, 16)};
                *reinterpret_cast<u32x2*>(p_ + 3 * X_PLANE) = u32x2{__builtin_amdgcn_alignbit(l1, l0, 16), __builtin_amdgcn_alignbit(l2, l1, 16)};
            }
        }
    };
    using std_true = std::integral_constant<bool, true>;
    using std_false = std::integral_constant<bool, false>;

    // producers: tile n + 1 goes into the other stage while the consumers multiply tile n; the loads of tile n + 2 are issued
    // right behind the stores and have a whole tile time to arrive.  One barrier per tile, for both roles.
    if (producer) {
        auto issue_t = [&](Stage& z, int N, int part) __attribute__((always_inline)) {
            if (is_edge(N)) issue(z, N, std_true{}, part); else issue(z, N, std_false{}, part);
        };
        auto store_t = [&](Stage& z, int N, int part) __attribute__((always_inline)) {
            unsigned char* const p_ = smem_w + ((N - n_lo) & 1) * BUF;
            if (N < n_hi) {
                if (is_edge(N)) store(z, N, p_, std_true{}, part); else store(z, N, p_, std_false{}, part);
            }
        };
        if constexpr (DEPTH == 1) {
            issue_t(st0, n_lo, 3);
            store_t(st0, n_lo, 3);
            issue_t(st0, n_lo + 1, 3);
            __syncthreads();
            // while tile n is multiplied: X of tile n + 1 goes into the other stage and the X loads of tile n + 2 go out, then the
            // same for gy -- every load is in flight over the other half's stores and the wait for the consumers
            for (int n = n_lo; n < n_hi; ++n) {
#if !(NEF_W2_DBG & 1)
                store_t(st0, n + 1, 2);
#endif
#if !(NEF_W2_DBG & 2)
                issue_t(st0, n + 2, 2);
#endif
#if !(NEF_W2_DBG & 1)
                store_t(st0, n + 1, 1);
#endif
#if !(NEF_W2_DBG & 2)
                issue_t(st0, n + 2, 1);
#endif
                __syncthreads();
            }
        } else {
            // two register sets: a tile's loads go out two tiles before its stores (a whole tile time in flight)
            issue_t(st0, n_lo, 3);
            issue_t(st1, n_lo + 1, 3);
            store_t(st0, n_lo, 3);
            issue_t(st0, n_lo + 2, 3);
            __syncthreads();
            for (int n = n_lo; n < n_hi; n += 2) {
#if !(NEF_W2_DBG & 1)
                store_t(st1, n + 1, 3);
#endif
#if !(NEF_W2_DBG & 2)
                issue_t(st1, n + 3, 3);
#endif
                __syncthreads();
                if (n + 1 < n_hi) {
#if !(NEF_W2_DBG & 1)
                    store_t(st0, n + 2, 3);
#endif
#if !(NEF_W2_DBG & 2)
                    issue_t(st0, n + 4, 3);
#endif
                    __syncthreads();
                }
            }
        }
    } else {
        __syncthreads();
        for (int n = n_lo; n < n_hi; ++n) {
            const unsigned char* const bufp = smem_w + ((n - n_lo) & 1) * BUF;
            const unsigned char* const ga = bufp + (wm * 32 + lo) * GP + hi * 16;
            const unsigned char* const xa = bufp + 2 * G_PLANE + (wn * 32 + lo) * XP + hi * 16;
#if !(NEF_W2_DBG & 4)
#pragma unroll
            for (int c = 0; c < NCH; ++c) h2w2_chunk<K, WR, GP, XP, G_PLANE, X_PLANE>(ga, xa, c, acc);
#endif
            __syncthreads();
        }
    }

    // this share's operand magnitudes: for the rescue, and for the call site's next launch
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        amax_x = fmaxf(amax_x, __shfl_xor(amax_x, o, 64));
        amax_g = fmaxf(amax_g, __shfl_xor(amax_g, o, 64));
    }
    if (lane == 0) Wl[wave_u] = amax_x, Wl[16 + wave_u] = amax_g;      // (consumer waves: 0)
    __syncthreads();
    {
        float wx = 0.f, wg_ = 0.f;
#pragma unroll
        for (int i = 0; i < NC + NP; ++i) wx = fmaxf(wx, Wl[i]), wg_ = fmaxf(wg_, Wl[16 + i]);
        __syncthreads();      // (the next pass stages over Wl)
        if (!last && !(wx * sx < 65000.f && wg_ * sg < 65000.f) && wx < 3e38f && wg_ < 3e38f) {
            rx_ = wx, rg_ = wg_;
            return false;
        }
    }
    if (producer) {
        if (a.clamped && lane == 0 && !(amax_x * sx < 65000.f && amax_g * sg < 65000.f)) atomicAdd(a.clamped, 1);      // not finite
        if (a.x_amax_next && lane == 0) {
            unsigned* const px = reinterpret_cast<unsigned*>(a.x_amax_next);
            unsigned* const pg = reinterpret_cast<unsigned*>(a.gy_amax_next);
            const unsigned bx = __builtin_bit_cast(unsigned, amax_x), bg = __builtin_bit_cast(unsigned, amax_g);
            if (amax_x < 3e38f && bx > __atomic_load_n(px, __ATOMIC_RELAXED)) atomicMax(px, bx);
            if (amax_g < 3e38f && bg > __atomic_load_n(pg, __ATOMIC_RELAXED)) atomicMax(pg, bg);
        }
        return true;
    }

    // partial sums: ws[split][g][k][co][ci]; a lane's column is ci = wn 32 + lo, its rows co = wm 32 + 4 hi + (r & 3) + 8 (r >> 2)
    const float ds = 1.f / (sx * sg);
    const int ci = ct * RX + wn * 32 + lo;
    float* const wsp = a.ws + ((int64_t)sp * a.G + g) * K * a.Cog * a.Cig;
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        const int co0 = mt * RG + (wm + i) * 32 + 4 * hi;
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) wsp[((int64_t)k * a.Cog + co0 + (r & 3) + 8 * (r >> 2)) * a.Cig + ci] = acc[i][k][r] * ds;
    }
    return true;
    };      // share_pass
    const float sx0 = scale_from(a.x_amax, a.x_scale), sg0 = scale_from(a.gy_amax, a.gy_scale);
    if (!share_pass(sx0, sg0, false))
        (void)share_pass(rx_ * sx0 < 65000.f ? sx0 : scale_for(rx_), rg_ * sg0 < 65000.f ? sg0 : scale_for(rg_), true);
}

template <int K, int PRO, int WM, int WN, int WR, int TTv, int NP, int DEPTH>
int launch_h2w2(const H2WArgs& a, hipStream_t st) {
    constexpr size_t lds = 2 * (2 * (32 * WM) * (TTv * 2 + 16) + 4 * (32 * WN) * ((TTv + 8) * 2));
    static unsigned long long lds_set = 0;
    if (int e = nef_ensure_dyn_lds(reinterpret_cast<const void*>(&conv_h2w2_kernel<K, PRO, WM, WN, WR, TTv, NP, DEPTH>), lds, &lds_set)) return e;
    const int members = a.m_tiles * a.c_tiles;
    const int teams8 = (a.teams + 7) / 8 * 8;
    const int64_t blocks = (int64_t)teams8 * members;
    if (blocks <= 0 || blocks > 0x7fffffff) return NEF_E_SHAPE;
    hipLaunchKernelGGL((conv_h2w2_kernel<K, PRO, WM, WN, WR, TTv, NP, DEPTH>), dim3((unsigned)blocks), dim3(64 * (WM / WR * WN + NP)), lds, st, a);
    return nef_launch_status();
}

// the second kernel takes the shapes with Cout_g % 128 == 0 (form 1: 128 x 64 channels per workgroup); NEF_H2W_V=1: the first kernel
// everywhere; NEF_H2W_64=<mask of prologue modes + 1>: form 2 (64 x 64 channels, 4 consumer + 8 producer waves) for the 64-channel
// layers with those prologues (bit p + 1 set: pro_mode p)
int h2w2_form(int Cog, int pro_mode) {
    static const bool v1 = nef_diag_env("NEF_H2W_V") && atoi(nef_diag_env("NEF_H2W_V")) == 1;
    static const int m64 = nef_diag_env("NEF_H2W_64") ? atoi(nef_diag_env("NEF_H2W_64")) : NEF_H2W_64_DEFAULT;
    if (v1) return 0;
    if (Cog % 128 == 0) return 1;
    return ((m64 >> pro_mode) & 1) ? 2 : 0;
}

}  // namespace

extern "C" {

__attribute__((visibility("hidden"))) bool nef_h2w_ok(int B, int T, int Cig, int Cog, int K, int pro_mode) {
    // pro_mode 4 / 5: bit 2 = clamped window ends (polyphase weight gradient), producer / consumer form only, no upsampling bit
    return (K == 1 || K == 3 || K == 7) && B > 0 && T >= TT && T % 2 == 0 && Cig % 64 == 0 && Cog % 64 == 0 && pro_mode >= 0 &&
           (pro_mode <= 3 || (pro_mode <= 5 && Cog % 128 == 0)) && (K == 3 || pro_mode == 0);
}

// team splits of the (sample, tile) sequence, and partial sums per (g, k, co, ci) the launch leaves (= splits, or twice that
// when the wave groups keep separate sums)
__attribute__((visibility("hidden"))) int nef_h2w_splits(int B, int T, int G, int Cig, int Cog, int K, int pro_mode, int* partials) {
    const int v2 = h2w2_form(Cog, pro_mode & 3);
    const H2WForm f = h2w_form(Cog, K);
    const int tps = (T + TT - 1) / TT;
    const int64_t n_tiles = (int64_t)B * tps;
    const int units = v2 ? G * (Cog / (v2 == 1 ? 128 : 64)) * (Cig / 64) : G * (Cog / (64 * f.mco)) * (Cig / 64);
    const int resident = v2 ? 1 : (f.split ? 1 : 2);      // workgroups per CU
    static const int rounds = nef_diag_env("NEF_H2W_ROUNDS") ? atoi(nef_diag_env("NEF_H2W_ROUNDS")) : 1;
    const int slots = rounds * resident * nef_cu_count();
    int S = slots / units;                 // `rounds` rounds of resident workgroups and never a workgroup more: one extra costs a whole round
    if (S > n_tiles) S = (int)n_tiles;
    if (S < 1) S = 1;
    if (partials) *partials = S * (!v2 && f.split == 2 ? 2 : 1);
    return S;
}

__attribute__((visibility("hidden"))) int nef_h2w_launch(const float* x, int64_t x_bs, int64_t x_gs, const float* in_scale, int64_t sc_bs,
                                                         int64_t sc_gs, const float* pro_a, const float* pro_b, int pro_mode, int pro_Bp,
                                                         const float* gy, int64_t gy_bs, int64_t gy_gs, float* ws, int B, int T, int G,
                                                         int Cig, int Cog, int K, int S, float x_scale, float gy_scale,
                                                         const float* x_amax, const float* gy_amax, float* x_amax_next,
                                                         float* gy_amax_next, int* clamped, hipStream_t st) {
    if (!nef_h2w_ok(B, T, Cig, Cog, K, pro_mode)) return NEF_E_SHAPE;
    if ((pro_mode & 1) && !(pro_a && pro_b && pro_Bp > 0)) return NEF_E_NULL;
    if ((x_amax_next == nullptr) != (gy_amax_next == nullptr)) return NEF_E_NULL;
    const H2WForm f = h2w_form(Cog, K);
    H2WArgs a;
    a.x = x, a.gy = gy, a.in_scale = in_scale, a.pro_a = pro_a, a.pro_b = pro_b, a.ws = ws;
    a.x_amax = x_amax, a.gy_amax = gy_amax, a.x_amax_next = x_amax_next, a.gy_amax_next = gy_amax_next, a.clamped = clamped;
    a.x_bs = x_bs, a.x_gs = x_gs, a.gy_bs = gy_bs, a.gy_gs = gy_gs, a.sc_bs = sc_bs, a.sc_gs = sc_gs;
    a.B = B, a.T = T, a.G = G, a.Cig = Cig, a.Cog = Cog, a.pro_Bp = pro_Bp > 0 ? pro_Bp : 1, a.S = S;
    a.tps = (T + TT - 1) / TT;
    a.n_tiles = B * a.tps;
    a.m_tiles = Cog / (64 * f.mco), a.c_tiles = Cig / 64;
    a.teams = G * S;
    a.x_scale = x_scale, a.gy_scale = gy_scale;
    a.xclamp = (pro_mode >> 2) & 1;
    pro_mode &= 3;
    const int Tin = (pro_mode & 2) ? T / 2 : T;
    a.x_end = ((int64_t)(B - 1) * x_bs + (int64_t)(G - 1) * x_gs + (int64_t)Cig * Tin) * 4;
    a.gy_end = ((int64_t)(B - 1) * gy_bs + (int64_t)(G - 1) * gy_gs + (int64_t)Cog * T) * 4;
    if (a.xclamp && !h2w2_form(Cog, pro_mode)) return NEF_E_UNSUPPORTED;      // (only the producer / consumer form continues the window)
    if (const int v2 = h2w2_form(Cog, pro_mode)) {
        a.m_tiles = Cog / (v2 == 1 ? 128 : 64), a.c_tiles = Cig / 64;
        // 12 waves = 168 registers each.  K = 7: 8 consumer waves of one row tile (7 x 16 accumulator registers) + 4 producer waves
        // with one staging set;  K = 3, 1: 4 consumer waves of two row tiles + 8 producer waves with two tiles in flight
#define NEF_H2W2(KK, PP)                                                                                             \
    (v2 == 1 ? launch_h2w2<KK, PP, 4, 2, (KK == 7 ? 1 : 2), 64, (KK == 7 ? 4 : 8), (KK == 7 ? 1 : 2)>(a, st)               \
             : launch_h2w2<KK, PP, 2, 2, 1, 64, 8, (KK == 7 ? 1 : 2)>(a, st))
        if (K == 7) return NEF_H2W2(7, 0);
        if (K == 1) return NEF_H2W2(1, 0);
        switch (pro_mode) {
            case 0: return NEF_H2W2(3, 0);
            case 1: return NEF_H2W2(3, 1);
            case 2: return NEF_H2W2(3, 2);
            default: return NEF_H2W2(3, 3);
        }
#undef NEF_H2W2
    }
    if (K == 7) return launch_h2w<7, 0, 1, 0>(a, st);
    if (K == 1) return launch_h2w<1, 0, 1, 0>(a, st);
    switch (pro_mode) {
        case 0: return launch_h2w<3, 0, 1, 0>(a, st);
        case 1: return launch_h2w<3, 1, 1, 0>(a, st);
        case 2: return launch_h2w<3, 2, 1, 0>(a, st);
        default: return launch_h2w<3, 3, 1, 0>(a, st);
    }
}

}  // extern "C"
