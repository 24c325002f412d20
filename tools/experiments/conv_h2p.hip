// conv_h2p.hip -- the split-fp16 direct convolution of conv_h2.hip (same arithmetic, same packed operand, same epilogue) as a
// PERSISTENT workgroup of PRODUCER and CONSUMER waves.
//
// conv_h2_kernel makes every wave do three jobs in turn -- fetch + split + stage its share of the activation tile, read
// fragments + issue matrix instructions, run the epilogue -- and its timing-only builds showed the two sides ADD by 15..30 %
// instead of overlapping (each side alone at the rate the chip sustains, the full kernel 15..30 % above the slower
// one).  Here the roles are separate waves of one 384-thread workgroup (two workgroups per CU, 168 registers per wave):
//   * waves 0..3, consumers: 2 (co) x 2 (t) waves on a 64-channel x 256-output tile, exactly conv_h2_kernel<K, PRO, 1>'s
//     arithmetic: B fragments by ds_read_b128 (ring over s = tap + t-tile), A fragments straight from L2 a tap ahead, three
//     v_mfma_f32_32x32x16_f16 per (tap, tile), then the epilogue (descale, bias / residual / ReLU / dropout / gate, BatchNorm
//     slot sums).  Nothing else: no address arithmetic of the activation tile, no conversions, no LDS stores;
//   * waves 4..5, producers: 8 input channels of a 16-channel stage each.  Loads go out TWO stages before their data is split
//     and stored (two register sets), so a load has two stage times to arrive whatever the consumers do; rows are fetched
//     through one buffer descriptor PER CHANNEL ROW (base = the row, extent = the row): positions left of the row (a negative
//     offset wraps) and right of it fail the hardware range check and read 0.0 -- the zero padding costs nothing and a lane's
//     offset is (t0 - PAD + lane) 4 in every tile, so the producers carry no per-tile vector state.
// One raw s_barrier per stage for both roles (a __syncthreads() would drain the producers' loads in flight).  The workgroup is
// persistent: it walks tiles id, id + grid, .. and the producers run ahead across the tile boundary -- while the consumers are
// in the epilogue of tile n, stage 0 of tile n + 1 is already being stored and stages 1, 2 are in flight, and the other
// workgroup of the CU keeps the matrix pipe busy.
//
// LDS image of a stage (fp16): [plane hi|lo][channel half 0|1][t mod 4][P4 = 66][8 channels] -- one 16-byte chunk per (position,
// channel half), chunk pitch 16 bytes: the consumers' fragment reads (lane = position) are 1 KB contiguous and conflict-free,
// and with P4 = 2 (mod 8) so are the producers' ds_write_b128 (8 consecutive lanes = positions r .. r + 7 = classes 0..3 x two
// neighbouring chunks each: chunk mod 8 all different).  (conv_h2_kernel's [position][16 channels] image makes every fragment
// read 2-way conflicted: a 16-lane group of ds_read_b128 covers 16 positions x 32 bytes = 512 bytes of address range for 256 of data.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "nefnet_hip.h"
#include "nef_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef NEF_H2P_T
#define NEF_H2P_T 0      // timing-only builds: 1 = producers issue no loads, 2 = no matrix instructions, 4 = no epilogue, 8 = producers do not split / store, 16 = A fragments fetched once per tile, 32 = B fragments read in the first stage only, 64 = producers read samples 0..7 only (real data, no HBM stream)
#endif

namespace {

// process-wide kernel-form options live in conv_h2.hip (nef_set_option / nef_get_option); this file only reads them
}  // namespace
__attribute__((visibility("hidden"))) int nef_opt_get(int key);
namespace {

constexpr int KC = 16;                 // input channels per stage
constexpr int NTO = 256;               // outputs per tile
constexpr int MT = 64;                 // output channels per tile
constexpr int NCW = 4, NPW = 2;        // consumer / producer waves of one GROUP
constexpr int NGR = 2;                 // groups per workgroup (each walks its own tiles; one hardware barrier for both)
constexpr int NTHREADS = 64 * (NCW + NPW) * NGR;
constexpr int P4 = 66;                 // chunks per (t mod 4) class
constexpr int HALF = 4 * P4 * 16;      // bytes of one 8-channel half of a plane
constexpr int PLANE = 2 * HALF;
constexpr int XBUF = 2 * PLANE;        // one stage: hi plane + lo plane
constexpr int ETAB = 6 * MT;           // floats of one epilogue table set
constexpr int RQ = 4;                  // output rows a lane finishes at a time in the epilogue (16 in all)
constexpr int EP = 16 / RQ;            // epilogue pieces = barrier intervals ("ticks") a group spends in its epilogue
constexpr int GRP_LDS = 2 * XBUF + 2 * ETAB * 4;      // bytes of LDS per group

__device__ __forceinline__ void h2p_split2s(float x0, float x1, float s, float lim, unsigned& h, unsigned& l) {
    x0 = __builtin_amdgcn_fmed3f(x0, -lim, lim);
    x1 = __builtin_amdgcn_fmed3f(x1, -lim, lim);
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(h) : "v"(x0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(h) : "v"(x1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(s), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(s), "v"(h));
}

// Tile walk.  Work id -> (xcd = id & 7, q = id >> 3); q -> (gm = q / (nb8 tps), sample group qb, tile of the sample qt): sample
// b0 = 8 qb + xcd, columns t0 = 256 qt -- every tile of a sample has the same id mod 8 (observed: the same XCD, so the halo columns
// two neighbouring tiles share are L2 hits) and neighbouring q (workgroups of the same round).  The grid is a multiple of 8, so a
// persistent workgroup (ids id0, id0 + grid, ..) stays on its XCD class and advances q by grid / 8: the walk is INCREMENTAL (adds
// and compares; the only divisions happen once per workgroup).  Samples 8 qb + xcd >= B (B not a multiple of 8) are empty tiles.
struct H2PWalk {
    int qt, qb, mt, g;
};
struct H2PGeom {
    int tps, nb8, m_tiles, G, B, xcd, qs_t, qs_b;
};

__device__ __forceinline__ H2PWalk h2p_walk_init(const H2PGeom& ge, int q0) {
    H2PWalk w;
    const int per = ge.nb8 * ge.tps;
    const int gm = q0 / per, qi = q0 - gm * per;
    w.qb = qi / ge.tps;
    w.qt = qi - w.qb * ge.tps;
    w.g = gm / ge.m_tiles;
    w.mt = gm - w.g * ge.m_tiles;
    return w;
}
__device__ __forceinline__ void h2p_walk_next(const H2PGeom& ge, H2PWalk& w) {
    w.qt += ge.qs_t;
    w.qb += ge.qs_b;
    if (w.qt >= ge.tps) w.qt -= ge.tps, ++w.qb;
    while (w.qb >= ge.nb8) {
        w.qb -= ge.nb8;
        if (++w.mt == ge.m_tiles) w.mt = 0, ++w.g;
    }
}
__device__ __forceinline__ bool h2p_walk_valid(const H2PGeom& ge, const H2PWalk& w) { return w.g < ge.G && 8 * w.qb + ge.xcd < ge.B; }

__device__ __forceinline__ void h2p_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int K, int PRO>
__global__ __launch_bounds__(NTHREADS, 3) void conv_h2p_kernel(nef_conv_args a_, int tps, int nb8, int m_tiles, int my_max) {
    // The arguments are read through the kernel-argument segment (a_ is its first member, offset 0) and the pointer is laundered
    // once per tile: a persistent loop otherwise keeps every field it will ever need live in scalar registers across the whole
    // tile loop (~90 of 102), spills them to vector lanes, and the vector registers those lanes cost spill to scratch in the
    // epilogue.  Re-reading a field is one scalar load from the constant cache.
    typedef const nef_conv_args __attribute__((address_space(4))) kargs_t;
    kargs_t* ap = (kargs_t*)__builtin_amdgcn_kernarg_segment_ptr();
#define a (*ap)
#define H2P_RELOAD_ARGS() asm volatile("" : "+s"(ap))
    constexpr bool UP = (PRO & 2) != 0, AFF = (PRO & 1) != 0;
    constexpr int PAD = (K - 1) / 2;
    constexpr int NSF = K + 3;                     // distinct B fragments per stage (s = tap + t-tile)
    constexpr int NM = UP ? 2 : 4, NS = UP ? 2 : 1;
    constexpr int NH = UP ? 8 : 4 * (K - 1);       // lanes that stage the right halo (positions 256 .. 256 + K - 2)
    static_assert(!UP || K == 3, "x2-upsampling prologue: K = 3");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_p[];
    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int group = wave_all / (NCW + NPW);                 // which of the workgroup's two tile streams this wave serves
    const int wave_u = wave_all - group * (NCW + NPW);        // 0 .. 3 consumers, 4 .. 5 producers
    unsigned char* const Xl = smem_p + group * GRP_LDS;                        // [2 buffers][XBUF]
    float* const El = reinterpret_cast<float*>(Xl + 2 * XBUF);                  // [2][ETAB] epilogue tables, by tile parity

    const int T = a.T, Cig = a.Cin_g, Cog = a.Cout_g;
    const int Tin = UP ? (T >> 1) : T;
    const int nst = Cig / KC;
    // Ticks.  One tick = one barrier interval of the workgroup.  A group spends nst ticks on the stages of a tile and EP ticks on
    // its epilogue, piece by piece; group 1 runs OFFT = (nst + EP) / 2 ticks behind group 0, so one group's epilogue ticks fall
    // beside the other group's stages: the matrix pipe always has a stream, and the stores of an epilogue leave under it.
    // Every group walks `my_max` tiles (those past the end of the work are empty): NT ticks for every wave of the workgroup.
    const int PER = nst + EP, OFFT = PER / 2;
    const int NT = my_max * PER + OFFT;
    H2PGeom ge;
    ge.tps = tps, ge.nb8 = nb8, ge.m_tiles = m_tiles, ge.G = a.G, ge.B = a.B, ge.xcd = (int)blockIdx.x & 7;
    {
        const int qs = (NGR * (int)gridDim.x) >> 3;           // group g of block b is stream b + g * grid of NGR * grid streams
        ge.qs_b = qs / tps;
        ge.qs_t = qs - ge.qs_b * tps;
    }
    const H2PWalk walk0 = h2p_walk_init(ge, ((int)blockIdx.x + group * (int)gridDim.x) >> 3);
    // input scale (an exact power of two, undone in the epilogue), as in conv_h2_kernel
    float xs_ = a.x_scale != 0.f ? a.x_scale : 1.f;
    if (a.x_amax) {
        const float m_ = a.x_amax[0];
        if (m_ > 0.f && m_ < 3e38f) {
            int e_;
            (void)frexpf(m_, &e_);
            xs_ = ldexpf(1.f, 9 - e_ < 100 ? 9 - e_ : 100);
        }
    }
    const float xlim_ = 65000.f / xs_;
    const _Float16* const wph = reinterpret_cast<const _Float16*>(a.wp);

    if (wave_u >= NCW) {
        // =========================================================================================== producers
        const int pw = wave_u - NCW;
        float amax_ = 0.f;
        // Staged position r <-> time t = t0 - PAD + r, r in [0, 256 + K - 1); its 16-byte chunk (8 channels of this wave's half) sits at
        // ((r & 3) P4 + (r >> 2)) 16.  Two lane -> position maps:
        //   FAST (tiles that touch neither end of the row): lane l owns the QUAD r = PAD + 4 l + j -- ONE 16-byte load per channel
        //        (8 loads per stage instead of 32: a wave's loads in flight stay far below the 63 the counter holds, so the two
        //        register sets really are two stages of prefetch; with dword loads the second set's issue blocked on the first) and
        //        no range checks at all; the K - 1 halo positions (PAD left, PAD right) go to lanes 0 .. 4 (K - 1) - 1, a channel
        //        pair each.  UP: lane l owns the sources m0 + 2 l .. + 2 (8-byte load + the one behind it) = the outputs r = 4 l + j.
        //   EDGE (first / last tile of a row): lane = position r = lane + 64 it (UP: interval lane + 64 it), every element checked
        //        by the hardware range check of a per-row descriptor or explicitly; the right halo to lanes 0 .. 4 (K - 1) - 1.
        const unsigned lbase = (unsigned)(lane * 16 + pw * HALF);
        const unsigned wofs = UP ? (unsigned)((2 * (lane & 1) * P4 + (lane >> 1)) * 16 + pw * HALF)
                                 : (unsigned)(((lane & 3) * P4 + (lane >> 2)) * 16 + pw * HALF);
        const int hcp = lane & 3, hp = lane >> 2;
        // EDGE right halo: position 256 + hp;  FAST halo: position hp (hp < PAD) or 256 + hp;  UP halo (both): position 256 + hp
        const unsigned hofs_e = (unsigned)((((hp & 3) * P4) + 64 + (hp >> 2)) * 16 + pw * HALF + hcp * 4);
        const unsigned hofs_f = UP ? hofs_e
                                   : (hp < PAD ? (unsigned)((hp * P4) * 16 + pw * HALF + hcp * 4) : hofs_e);

        struct Set {
            float xv[32];
            float xh[2][NS];
        };
        Set S0, S1;
        // a stream = a position of the (tile, stage) walk: the loads run two stages ahead of the stores
        struct Stream {
            H2PWalk w;
            int st, t0;
            bool valid, edge;
            const float* rowb;       // row 0 of the tile's sample and group
            const float* pa;         // AFF: pro_a (pb = pa + pb_off) of the tile's pass and group;  else: in_scale of the sample and group
        };
        const int64_t pb_off = AFF ? (a.pro_b - a.pro_a) : 0;
        auto enter = [&](Stream& s) __attribute__((always_inline)) {
            H2P_RELOAD_ARGS();
            s.valid = h2p_walk_valid(ge, s.w);
            const int b0 = s.valid ? 8 * s.w.qb + ge.xcd : 0, g = s.valid ? s.w.g : 0;
            s.t0 = s.w.qt * NTO;
            s.edge = UP ? !(s.t0 > 0 && s.t0 + NTO + 2 <= T) : !(s.t0 > 0 && s.t0 + NTO + PAD <= T);
            s.rowb = a.x + (int64_t)((NEF_H2P_T & 64) ? (b0 & 7) : b0) * a.x_bs + (int64_t)g * a.x_gs;      // (64: the same 8 samples for every tile: L2 hits)
            if constexpr (AFF) s.pa = a.pro_a + ((b0 / a.pro_Bp) * a.G * Cig + g * Cig);
            else s.pa = a.in_scale ? a.in_scale + (int64_t)b0 * a.sc_bs + (int64_t)g * a.sc_gs : nullptr;
        };
        auto adv = [&](Stream& s) __attribute__((always_inline)) {
            if (++s.st == nst) {
                s.st = 0;
                h2p_walk_next(ge, s.w);
                enter(s);
            }
        };

        auto issue = [&](Set& z, const Stream& s, auto edge_c) __attribute__((always_inline)) {
            constexpr bool EDGE = decltype(edge_c)::value;
#if !(NEF_H2P_T & 1)
            const float* const rowp = s.rowb + (int64_t)(s.st * KC + 8 * pw) * Tin;
            const unsigned ext = s.valid ? (unsigned)(Tin * 4) : 0u;
            if constexpr (!EDGE) {
                if constexpr (UP) {
                    const unsigned vo = (unsigned)(((s.t0 >> 1) - 1 + 2 * lane) * 4);
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const __amdgpu_buffer_rsrc_t rs = nef_rsrc_n(rowp + (int64_t)c * Tin, ext);
                        const nef_f32x2 v2 = nef_buf_f32x2(rs, vo, 0);
                        z.xv[3 * c] = v2[0], z.xv[3 * c + 1] = v2[1];
                        z.xv[3 * c + 2] = nef_buf_f32(rs, vo + 8u, 0);
                    }
                } else {
                    const unsigned vo = (unsigned)((s.t0 + 4 * lane) * 4);
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const __amdgpu_buffer_rsrc_t rs = nef_rsrc_n(rowp + (int64_t)c * Tin, ext);
                        const nef_f32x4 v4 = nef_buf_f32x4(rs, vo, 0);
                        z.xv[4 * c] = v4[0], z.xv[4 * c + 1] = v4[1], z.xv[4 * c + 2] = v4[2], z.xv[4 * c + 3] = v4[3];
                    }
                }
            } else {
                unsigned vo[NM][NS];
#pragma unroll
                for (int it = 0; it < NM; ++it) {
                    if constexpr (UP) {
                        const int m = (s.t0 >> 1) - 1 + lane + 64 * it;
                        const int mb = m + 1 < Tin ? m + 1 : Tin - 1;
                        vo[it][0] = (unsigned)(m * 4);           // m = -1 wraps: out of range, reads 0.0
                        vo[it][NS - 1] = (unsigned)(mb * 4);
                    } else {
                        vo[it][0] = (unsigned)((s.t0 - PAD + lane + 64 * it) * 4);
                    }
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const __amdgpu_buffer_rsrc_t rs = nef_rsrc_n(rowp + (int64_t)c * Tin, ext);
#pragma unroll
                    for (int it = 0; it < NM; ++it)
#pragma unroll
                        for (int ns = 0; ns < NS; ++ns) z.xv[(it * 8 + c) * NS + ns] = nef_buf_f32(rs, vo[it][ns], 0);
                }
            }
            if constexpr (K > 1) {
                const __amdgpu_buffer_rsrc_t rsS = nef_rsrc_n(rowp, s.valid ? (unsigned)(8 * Tin * 4) : 0u);
                if constexpr (UP) {
                    const int m = (s.t0 >> 1) - 1 + 128;
                    const int mb = m + 1 < Tin ? m + 1 : Tin - 1;
                    const bool ok = lane < NH && m < Tin;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        z.xh[j][0] = nef_buf_f32(rsS, ok ? (unsigned)(((2 * hcp + j) * Tin + m) * 4) : NEF_OOB, 0);
                        z.xh[j][NS - 1] = nef_buf_f32(rsS, ok ? (unsigned)(((2 * hcp + j) * Tin + mb) * 4) : NEF_OOB, 0);
                    }
                } else {
                    const int r = (!EDGE && hp < PAD) ? hp : 256 + hp;
                    const int t = s.t0 - PAD + r;
                    const bool ok = lane < NH && t >= 0 && t < T;
#pragma unroll
                    for (int j = 0; j < 2; ++j) z.xh[j][0] = nef_buf_f32(rsS, ok ? (unsigned)(((2 * hcp + j) * Tin + t) * 4) : NEF_OOB, 0);
                }
            }
#endif
        };

        // SC: the launch has a per-channel input factor (in_scale; PRO == 0 only)
        auto store = [&](Set& z, const Stream& s, unsigned char* bufp, auto sc_c, auto edge_c) __attribute__((always_inline)) {
            constexpr bool SC = decltype(sc_c)::value;
            constexpr bool EDGE = decltype(edge_c)::value;
#if !(NEF_H2P_T & 8)
            const int c0 = s.st * KC + 8 * pw;
            float pa[8], pb[8], ha[2], hb[2];      // channel parameters: wave-uniform (scalar loads); a halo lane's own pair
            if constexpr (AFF || SC) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    pa[c] = s.pa[c0 + c];
                    if constexpr (AFF) pb[c] = s.pa[pb_off + c0 + c];
                }
                if constexpr (K > 1) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        ha[j] = s.pa[c0 + 2 * hcp + j];
                        if constexpr (AFF) hb[j] = s.pa[pb_off + c0 + 2 * hcp + j];
                    }
                }
            }
            // 8 channels of one position -> its hi and lo chunks
            auto put = [&](const float (&v)[8], unsigned char* p_) __attribute__((always_inline)) {
                u32x4 h, l;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned h_, l_;
                    h2p_split2s(v[2 * q], v[2 * q + 1], xs_, xlim_, h_, l_);
                    h[q] = h_, l[q] = l_;
                }
                *reinterpret_cast<u32x4*>(p_) = h;
                *reinterpret_cast<u32x4*>(p_ + PLANE) = l;
            };
            if constexpr (UP) {
                if constexpr (!EDGE) {
                    float sv[8][3];
#pragma unroll
                    for (int c = 0; c < 8; ++c)
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            float x_ = z.xv[3 * c + i];
                            if constexpr (AFF) x_ = fmaxf(fmaf(x_, pa[c], pb[c]), 0.f);
                            sv[c][i] = x_;
                        }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {      // position r = 4 l + j: interval (s[j / 2], s[j / 2 + 1]), weights .75 / .25 (j even) or .25 / .75
                        float v[8];
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const float sa = sv[c][j >> 1], sb = sv[c][(j >> 1) + 1];
                            const float o = (j & 1) ? (1.f - 0.75f) * sa + 0.75f * sb : (1.f - 0.25f) * sa + 0.25f * sb;
                            amax_ = fmaxf(amax_, fabsf(o));
                            v[c] = o;
                        }
                        put(v, bufp + lbase + (j * P4) * 16);
                    }
                } else {
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int m = (s.t0 >> 1) - 1 + lane + 64 * it;
                        const bool ok0 = (unsigned)(2 * m + 1) < (unsigned)T, ok1 = (unsigned)(2 * m + 2) < (unsigned)T;
                        const float ulam = m == -1 ? 1.f : 0.75f;
                        float v0[8], v1[8];
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            float sa = z.xv[(it * 8 + c) * NS], sb = z.xv[(it * 8 + c) * NS + NS - 1];
                            if constexpr (AFF) sa = fmaxf(fmaf(sa, pa[c], pb[c]), 0.f), sb = fmaxf(fmaf(sb, pa[c], pb[c]), 0.f);
                            float o0 = (1.f - 0.25f) * sa + 0.25f * sb;
                            float o1 = (1.f - ulam) * sa + ulam * sb;
                            o0 = ok0 ? o0 : 0.f;
                            o1 = ok1 ? o1 : 0.f;
                            amax_ = fmaxf(amax_, fmaxf(fabsf(o0), fabsf(o1)));
                            v0[c] = o0, v1[c] = o1;
                        }
                        put(v0, bufp + wofs + it * (32 * 16));
                        put(v1, bufp + wofs + it * (32 * 16) + P4 * 16);
                    }
                }
                if (lane < NH) {      // the right halo interval: position 256 + hp of the channel pair 2 hcp, 2 hcp + 1
                    const int m = (s.t0 >> 1) - 1 + 128;
                    const bool ok = (unsigned)(2 * m + 1 + hp) < (unsigned)T;
                    float o[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float sa = z.xh[j][0], sb = z.xh[j][NS - 1];
                        if constexpr (AFF) sa = fmaxf(fmaf(sa, ha[j], hb[j]), 0.f), sb = fmaxf(fmaf(sb, ha[j], hb[j]), 0.f);
                        const float oa = (1.f - 0.25f) * sa + 0.25f * sb;
                        const float ob = (1.f - 0.75f) * sa + 0.75f * sb;
                        o[j] = ok ? (hp == 0 ? oa : ob) : 0.f;
                    }
                    amax_ = fmaxf(amax_, fmaxf(fabsf(o[0]), fabsf(o[1])));
                    unsigned h_, l_;
                    h2p_split2s(o[0], o[1], xs_, xlim_, h_, l_);
                    *reinterpret_cast<unsigned*>(bufp + hofs_e) = h_;
                    *reinterpret_cast<unsigned*>(bufp + hofs_e + PLANE) = l_;
                }
            } else {
                if constexpr (!EDGE) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {      // position r = PAD + 4 l + j
                        float v[8];
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            float x_ = z.xv[4 * c + j];
                            if constexpr (AFF) x_ = fmaxf(fmaf(x_, pa[c], pb[c]), 0.f);
                            if constexpr (SC) x_ *= pa[c];
                            amax_ = fmaxf(amax_, fabsf(x_));
                            v[c] = x_;
                        }
                        put(v, bufp + lbase + ((((PAD + j) & 3) * P4) + ((PAD + j) >> 2)) * 16);
                    }
                } else {
#pragma unroll
                    for (int it = 0; it < NM; ++it) {
                        const bool ok = (unsigned)(s.t0 - PAD + lane + 64 * it) < (unsigned)T;
                        float v[8];
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            float x_ = z.xv[(it * 8 + c) * NS];
                            if constexpr (AFF) x_ = fmaxf(fmaf(x_, pa[c], pb[c]), 0.f);
                            if constexpr (PRO != 0) x_ = ok ? x_ : 0.f;
                            if constexpr (SC) x_ *= pa[c];
                            amax_ = fmaxf(amax_, fabsf(x_));
                            v[c] = x_;
                        }
                        put(v, bufp + wofs + it * (16 * 16));
                    }
                }
                if constexpr (K > 1) {
                    if (lane < NH) {
                        const int r = (!EDGE && hp < PAD) ? hp : 256 + hp;
                        const int t = s.t0 - PAD + r;
                        const bool ok = t >= 0 && t < T;
                        float o[2];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            float x_ = z.xh[j][0];
                            if constexpr (AFF) x_ = fmaxf(fmaf(x_, ha[j], hb[j]), 0.f);
                            if constexpr (PRO != 0) x_ = ok ? x_ : 0.f;
                            if constexpr (SC) x_ *= ha[j];
                            o[j] = x_;
                        }
                        amax_ = fmaxf(amax_, fmaxf(fabsf(o[0]), fabsf(o[1])));
                        unsigned h_, l_;
                        h2p_split2s(o[0], o[1], xs_, xlim_, h_, l_);
                        const unsigned ho = EDGE ? hofs_e : hofs_f;
                        *reinterpret_cast<unsigned*>(bufp + ho) = h_;
                        *reinterpret_cast<unsigned*>(bufp + ho + PLANE) = l_;
                    }
                }
            }
#endif
        };
        const bool has_sc = PRO == 0 && a.in_scale != nullptr;
        auto issue_v = [&](Set& z, const Stream& s) __attribute__((always_inline)) {
            if (s.edge) issue(z, s, std::true_type{});
            else issue(z, s, std::false_type{});
        };
        auto store_e = [&](Set& z, const Stream& s, unsigned char* bufp, auto sc_c) __attribute__((always_inline)) {
            if (s.edge) store(z, s, bufp, sc_c, std::true_type{});
            else store(z, s, bufp, sc_c, std::false_type{});
        };
        auto store_v = [&](Set& z, const Stream& s, unsigned char* bufp) __attribute__((always_inline)) {
            if (!s.valid) return;
            if constexpr (PRO == 0) {
                if (has_sc) store_e(z, s, bufp, std::true_type{});
                else store_e(z, s, bufp, std::false_type{});
            } else {
                store_e(z, s, bufp, std::false_type{});
            }
        };

        Stream sI, sS;
        sI.w = walk0, sI.st = 0;
        enter(sI);
        sS = sI;
        issue_v(S0, sI);
        adv(sI);
        issue_v(S1, sI);
        adv(sI);
        store_v(S0, sS, Xl);
        adv(sS);
        issue_v(S0, sI);
        adv(sI);
        h2p_barrier();
        // Stage s of the group's stream lives in register set s & 1 and goes to LDS buffer s & 1.  Before a tick, stages 0 .. S - 1 are
        // stored and 0 .. C - 1 consumed; during the tick the consumers read buffer C & 1 (a stage tick) or nothing (an epilogue or
        // idle tick), so stage S may be stored iff S <= C + 1 -- one store per tick at most, behind it the loads of stage S + 2.
        int S = 1, C = 0;
        int ph = -group * OFFT;          // position in the group's period: [0, nst) stages, [nst, PER) epilogue pieces; < 0: not started
        int tiles_left = my_max;
        for (int k = 0; k < NT; ++k) {
            const bool stage_tick = ph >= 0 && ph < nst && tiles_left > 0;
            if (S <= C + 1) {
                if (S & 1) {
                    store_v(S1, sS, Xl + XBUF);
                    adv(sS);
                    issue_v(S1, sI);
                    adv(sI);
                } else {
                    store_v(S0, sS, Xl);
                    adv(sS);
                    issue_v(S0, sI);
                    adv(sI);
                }
                ++S;
            }
            if (stage_tick) ++C;
            if (++ph == PER) ph = 0, --tiles_left;
            h2p_barrier();
        }
        // this launch's own input magnitude, for the call site's next launch
        H2P_RELOAD_ARGS();
        if (a.x_amax_next || a.x_clamped) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) amax_ = fmaxf(amax_, __shfl_xor(amax_, o, 64));
            if (a.x_clamped && lane == 0 && !(amax_ * xs_ < 65000.f)) atomicAdd(a.x_clamped, 1);
        }
        if (a.x_amax_next) {
            if (lane == 0 && amax_ < 3e38f) {
                unsigned* const p_ = reinterpret_cast<unsigned*>(a.x_amax_next);
                const unsigned b_ = __builtin_bit_cast(unsigned, amax_);
                if (b_ > __atomic_load_n(p_, __ATOMIC_RELAXED)) atomicMax(p_, b_);
            }
        }
        return;
    }

    // =============================================================================================== consumers
    const int lo = lane & 31, hi = lane >> 5;
    const int wm = wave_u >> 1, wn = wave_u & 1;
    const int ncot = Cog / 32, nc16 = Cig / 16;
    const unsigned a_tap = (unsigned)(ncot * 2 * 1024);      // bytes between taps of one chunk
    const unsigned avo = (unsigned)(lane * 16);
    const unsigned fb_lane = (unsigned)(hi * HALF + (wn * 32 + lo) * 16);
    const int64_t ctot = (int64_t)a.G * Cog;
    float* const slot_out = a.bnb_slots ? a.bnb_slots : a.stats;
    const float* const dsc = reinterpret_cast<const float*>(wph + (int64_t)a.G * K * Cog * Cig * 2);

    h2p_barrier();
    for (int i = 0; i < group * OFFT; ++i) h2p_barrier();      // group 1 starts OFFT ticks late
    int gs = 0;
    H2PWalk cw = walk0;
    for (int ti = 0; ti < my_max; ++ti) {
        H2P_RELOAD_ARGS();
        const bool tile_ok = h2p_walk_valid(ge, cw);
        const int b0 = tile_ok ? 8 * cw.qb + ge.xcd : 0, t0 = cw.qt * NTO, m0 = cw.mt * MT, g = tile_ok ? cw.g : 0;      // an empty tile: addresses of sample 0 (the optional operands are fetched unconditionally), nothing is stored
        h2p_walk_next(ge, cw);
        float* const E = El + (ti & 1) * ETAB;
        // the tile's epilogue tables (wave 0; read after the tile's last barrier, and E[ti & 1] was last read in the epilogue of tile
        // ti - 2, which every wave left before the first barrier of tile ti - 1)
        float et[6];
        if (wave_u == 0) {      // fetched here, stored to LDS behind the first stage
            const int ch_ = g * Cog + m0 + lane;
            et[0] = a.bias ? a.bias[ch_] : 0.f;
            et[5] = dsc[ch_];
            if (a.bnb_slots) {
                const int pr_ = (b0 / a.bnb_Bp) * a.G * Cog + ch_;
                et[1] = a.bnb_mean[pr_];
                et[2] = a.bnb_invstd[pr_];
                et[3] = a.bnb_a[pr_];
                et[4] = a.bnb_b[pr_];
            }
        }
        // this wave's 2 A fragments of (chunk c, tap kk): ((g nc16 + c) K + kk) ncot 2 KB + (m0 / 32 + wm) 2 KB
        const __amdgpu_buffer_rsrc_t wrs = nef_rsrc(wph + ((int64_t)g * nc16 * K * ncot * 2 + (int64_t)(m0 / 32 + wm) * 2) * 512);

        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        h16x8 fa[2][2];             // [set][plane]
#define H2P_A_ISSUE(CH, KK, SET)                                                                                      \
    {                                                                                                               \
        const unsigned so_ = (unsigned)(((CH) * K + (KK)) * a_tap);                                                 \
        if (!(NEF_H2P_T & 16) || (CH) + (KK) == 0) {                                                                \
            fa[SET][0] = __builtin_bit_cast(h16x8, nef_buf_f32x4(wrs, avo, so_));                                   \
            fa[SET][1] = __builtin_bit_cast(h16x8, nef_buf_f32x4(wrs, avo, so_ + 1024u));                           \
        }                                                                                                           \
    }
        H2P_A_ISSUE(0, 0, 0)

        // one stage; PAR = parity of the stage within the tile (the A ring's phase: tap kk of the stage sits in set (PAR K + kk) & 1)
        auto stage = [&](int st, auto par_c) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par_c)::value;
            const unsigned char* const xb = Xl + (gs & 1) * XBUF + fb_lane;
            const bool more = st + 1 < nst;
            h16x8 fb[5][2];              // ring over s: [slot][plane]
#define H2P_B_LOAD(S)                                                                                                \
    {                                                                                                               \
        const unsigned char* p_ = xb + ((((S) & 3) * P4 + ((S) >> 2)) * 16);                                        \
        if (!(NEF_H2P_T & 32) || st == 0) {                                                                         \
            fb[(S) % 5][0] = *reinterpret_cast<const h16x8*>(p_);                                                   \
            fb[(S) % 5][1] = *reinterpret_cast<const h16x8*>(p_ + PLANE);                                           \
        }                                                                                                           \
    }
            H2P_B_LOAD(0)
            H2P_B_LOAD(1)
            H2P_B_LOAD(2)
            H2P_B_LOAD(3)
#pragma unroll
            for (int kk = 0; kk < K; ++kk) {
                const int s_ = (PAR * K + kk) & 1;
                // next tap's A fragments (the next stage's first tap behind the last one; past the tile's end: a repeat, harmless)
                if (kk + 1 < K) H2P_A_ISSUE(st, kk + 1, s_ ^ 1)
                else H2P_A_ISSUE(more ? st + 1 : st, 0, s_ ^ 1)
                if (kk + 4 < NSF) H2P_B_LOAD(kk + 4)
                __builtin_amdgcn_s_setprio(1);
#if !(NEF_H2P_T & 2)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s_][0], fb[(kk + j) % 5][0], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s_][0], fb[(kk + j) % 5][1], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s_][1], fb[(kk + j) % 5][0], acc[j], 0, 0, 0);
#endif
            }
#undef H2P_B_LOAD
            ++gs;
            h2p_barrier();
        };
        stage(0, std::integral_constant<int, 0>{});
        if (wave_u == 0) {
            E[lane] = et[0];
            E[5 * MT + lane] = et[5] / xs_;
            if (a.bnb_slots) {
                E[MT + lane] = et[1];
                E[2 * MT + lane] = et[2];
                E[3 * MT + lane] = et[3];
                E[4 * MT + lane] = et[4];
            }
        }
        stage(1, std::integral_constant<int, 1>{});
        for (int st = 2; st < nst; st += 2) {
            stage(st, std::integral_constant<int, 0>{});
            if (st + 1 < nst) stage(st + 1, std::integral_constant<int, 1>{});
        }
#undef H2P_A_ISSUE

#if NEF_H2P_T & 4
        {
            float z_ = 0.f;
            for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) z_ += acc[j][r];
            if (z_ == 12345.678f) a.y[0] = z_;
            for (int i = 0; i < EP; ++i) h2p_barrier();
            continue;
        }
#endif
        // ---- epilogue (conv_h2_kernel's, TM = 1): descale, then bias / residual / ReLU / dropout / gate on the four adjacent outputs
        // a lane owns per row
        const int t = t0 + wn * 128 + 4 * lo;
        const bool inb = tile_ok;
        const bool live[2] = {inb && t < T, inb && t + 2 < T};
        const int ts[2] = {live[0] ? t : 0, live[1] ? t + 2 : 0};
        const bool ragged = t0 + NTO > T;     // workgroup-uniform
        const int cobase = m0 + wm * 32 + 4 * hi;
        const int erow0 = wm * 32 + 4 * hi;
        float sv[32];
#pragma unroll
        for (int h = 0; h < 16 / RQ; ++h) {      // RQ rows of the wave's 16 (per 32-row tile half `hi`) at a time
#define NEF_ROW(q) ((((q) + RQ * h) & 3) + 8 * (((q) + RQ * h) >> 2))
            float y[RQ][4];
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                const float ds = E[5 * MT + erow0 + NEF_ROW(q)];
                const float bv = E[erow0 + NEF_ROW(q)];
#pragma unroll
                for (int e = 0; e < 4; ++e) y[q][e] = fmaf(acc[e][q + RQ * h], ds, bv);
            }
#define NEF_EPI_FETCH4(PTR, BS, GS, DST)                                                                              \
    if (!ragged) {                                                                                                  \
        const __amdgpu_buffer_rsrc_t rs_ =                                                                          \
            nef_rsrc((PTR) + (int64_t)b0 * (BS) + (int64_t)g * (GS) + (int64_t)(m0 + wm * 32) * T);                  \
        const unsigned vo_ = inb ? (unsigned)((4 * hi * T + t) * 4) : NEF_OOB;                                      \
        _Pragma("unroll") for (int q = 0; q < RQ; ++q) {                                                             \
            const f32x4 t4 = nef_buf_f32x4(rs_, vo_, (unsigned)(NEF_ROW(q) * T * 4));                               \
            DST[q][0] = t4[0]; DST[q][1] = t4[1]; DST[q][2] = t4[2]; DST[q][3] = t4[3];                             \
        }                                                                                                           \
    } else {                                                                                                        \
        _Pragma("unroll") for (int pr = 0; pr < 2; ++pr) {                                                          \
            const float* p_ = (PTR) + (int64_t)b0 * (BS) + (int64_t)g * (GS) + (int64_t)cobase * T + ts[pr];        \
            _Pragma("unroll") for (int q = 0; q < RQ; ++q) {                                                         \
                const f32x2 t2 = *reinterpret_cast<const f32x2*>(p_ + (int64_t)NEF_ROW(q) * T);                     \
                DST[q][2 * pr] = t2[0];                                                                             \
                DST[q][2 * pr + 1] = t2[1];                                                                         \
            }                                                                                                       \
        }                                                                                                           \
    }
            if (a.res) {
                float rv[RQ][4];
                NEF_EPI_FETCH4(a.res, a.res_bs, a.res_gs, rv)
#pragma unroll
                for (int q = 0; q < RQ; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[q][e] += rv[q][e];
            }
            if (a.relu) {
#pragma unroll
                for (int q = 0; q < RQ; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[q][e] = fmaxf(y[q][e], 0.f);
            }
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {        // dropout works on the two output pairs (t, t+1), (t+2, t+3)
                if (a.mask) {
                    const uint8_t* mp = a.mask + ((int64_t)b0 * ctot + (int64_t)g * Cog + cobase) * T + ts[pr];
                    unsigned short t8[RQ];
#pragma unroll
                    for (int q = 0; q < RQ; ++q) t8[q] = *reinterpret_cast<const unsigned short*>(mp + (int64_t)NEF_ROW(q) * T);
#pragma unroll
                    for (int q = 0; q < RQ; ++q) {
                        y[q][2 * pr] *= (float)(t8[q] & 0xff) * a.drop_scale;
                        y[q][2 * pr + 1] *= (float)(t8[q] >> 8) * a.drop_scale;
                    }
                } else if (a.drop_p > 0.f) {
                    const int64_t d0 = ((int64_t)b0 * ctot + (int64_t)g * Cog + cobase) * T + ts[pr];
                    const uint64_t seed = a.rng_seed + (a.rng_seed_dev ? a.rng_seed_dev[0] : 0ull);
#pragma unroll
                    for (int q = 0; q < RQ; ++q) {
                        const uint64_t dense = (uint64_t)(d0 + (int64_t)NEF_ROW(q) * T);
                        float u0, u1;
                        nef_rng_uniform2(seed, dense, u0, u1);
                        y[q][2 * pr] = (u0 >= a.drop_p) ? y[q][2 * pr] * a.drop_scale : 0.f;
                        y[q][2 * pr + 1] = (u1 >= a.drop_p) ? y[q][2 * pr + 1] * a.drop_scale : 0.f;
                    }
                }
            }
            if (a.gate) {
                float gv[RQ][4];
                NEF_EPI_FETCH4(a.gate, a.gate_bs, a.gate_gs, gv)
#pragma unroll
                for (int q = 0; q < RQ; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[q][e] = gv[q][e] > 0.f ? y[q][e] * a.gate_scale : 0.f;
            }
#undef NEF_EPI_FETCH4
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                if (live[pr] && !live[1]) {      // the half-live quad at the end of a row with T % 4 == 2
                    float* yp = a.y + (int64_t)b0 * a.y_bs + (int64_t)g * a.y_gs + (int64_t)cobase * T + t + 2 * pr;
#pragma unroll
                    for (int q = 0; q < RQ; ++q) {
                        f32x2 o;
                        o[0] = y[q][2 * pr];
                        o[1] = y[q][2 * pr + 1];
                        *reinterpret_cast<f32x2*>(yp + (int64_t)NEF_ROW(q) * T) = o;
                    }
                }
            }
            {
                const __amdgpu_buffer_rsrc_t yrs =
                    nef_rsrc(a.y + (int64_t)b0 * a.y_bs + (int64_t)g * a.y_gs + (int64_t)(m0 + wm * 32) * T);
                const unsigned yvo = live[1] ? (unsigned)((4 * hi * T + t) * 4) : NEF_OOB;
#pragma unroll
                for (int q = 0; q < RQ; ++q) {
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
                const bool interior = j2 >= 1 && j2 + 2 < Lh;
                auto sums = [&](int q, float xa, float xb_, float xc, float xd) __attribute__((always_inline)) {
                    const int er = erow0 + NEF_ROW(q);
                    const float af = E[3 * MT + er], bf = E[4 * MT + er];
                    const float mf = E[MT + er], is = E[2 * MT + er];
                    const float ma = fmaf(xa, af, bf) > 0.f ? 1.f : 0.f, mb = fmaf(xb_, af, bf) > 0.f ? 1.f : 0.f;
                    const float mc = fmaf(xc, af, bf) > 0.f ? 1.f : 0.f, md = fmaf(xd, af, bf) > 0.f ? 1.f : 0.f;
                    const float ha = ma * ((xa - mf) * is), hb = mb * ((xb_ - mf) * is);
                    const float hc = mc * ((xc - mf) * is), hd = md * ((xd - mf) * is);
                    const float g0 = live[0] ? y[q][0] : 0.f, g1 = live[0] ? y[q][1] : 0.f;
                    const float g2 = live[1] ? y[q][2] : 0.f, g3 = live[1] ? y[q][3] : 0.f;
                    sv[2 * (q + RQ * h)] = fmaf(g0, fmaf(0.75f, mb, 0.25f * ma), g1 * fmaf(0.75f, mb, 0.25f * mc)) +
                                          fmaf(g2, fmaf(0.75f, mc, 0.25f * mb), g3 * fmaf(0.75f, mc, 0.25f * md));
                    sv[2 * (q + RQ * h) + 1] = fmaf(g0, fmaf(0.75f, hb, 0.25f * ha), g1 * fmaf(0.75f, hb, 0.25f * hc)) +
                                              fmaf(g2, fmaf(0.75f, hc, 0.25f * hb), g3 * fmaf(0.75f, hc, 0.25f * hd));
                };
                if (interior) {
                    f32x4_a4 xv[RQ];
#pragma unroll
                    for (int q = 0; q < RQ; ++q) xv[q] = *reinterpret_cast<const f32x4_a4*>(xp + (int64_t)NEF_ROW(q) * Lh + (j2 - 1));
#pragma unroll
                    for (int q = 0; q < RQ; ++q) sums(q, xv[q][0], xv[q][1], xv[q][2], xv[q][3]);
                } else {
#pragma unroll
                    for (int q = 0; q < RQ; ++q) {
                        const float* xr = xp + (int64_t)NEF_ROW(q) * Lh;
                        sums(q, xr[im1], xr[j2], xr[i1], xr[ip2]);
                    }
                }
            } else if (a.bnb_slots) {
                const float* xp = a.bnb_x + ((int64_t)b0 * ctot + (int64_t)g * Cog + cobase) * T;
#pragma unroll
                for (int q = 0; q < RQ; ++q) {
                    const int row = NEF_ROW(q);
                    const int er = erow0 + row;
                    const float af = E[3 * MT + er], bf = E[4 * MT + er];
                    const float mf = E[MT + er], is = E[2 * MT + er];
                    const f32x2 x01 = *reinterpret_cast<const f32x2*>(xp + (int64_t)row * T + ts[0]);
                    const f32x2 x23 = *reinterpret_cast<const f32x2*>(xp + (int64_t)row * T + ts[1]);
                    const float g0 = (live[0] && fmaf(x01[0], af, bf) > 0.f) ? y[q][0] : 0.f;
                    const float g1 = (live[0] && fmaf(x01[1], af, bf) > 0.f) ? y[q][1] : 0.f;
                    const float g2 = (live[1] && fmaf(x23[0], af, bf) > 0.f) ? y[q][2] : 0.f;
                    const float g3 = (live[1] && fmaf(x23[1], af, bf) > 0.f) ? y[q][3] : 0.f;
                    sv[2 * (q + RQ * h)] = (g0 + g1) + (g2 + g3);
                    sv[2 * (q + RQ * h) + 1] = fmaf(g0, (x01[0] - mf) * is, g1 * ((x01[1] - mf) * is)) +
                                              fmaf(g2, (x23[0] - mf) * is, g3 * ((x23[1] - mf) * is));
                }
            } else if (a.stats) {
#pragma unroll
                for (int q = 0; q < RQ; ++q) {
                    const float y0 = live[0] ? y[q][0] : 0.f, y1 = live[0] ? y[q][1] : 0.f;
                    const float y2 = live[1] ? y[q][2] : 0.f, y3 = live[1] ? y[q][3] : 0.f;
                    sv[2 * (q + RQ * h)] = (y0 + y1) + (y2 + y3);
                    sv[2 * (q + RQ * h) + 1] = fmaf(y0, y0, y1 * y1) + fmaf(y2, y2, y3 * y3);
                }
            }
#undef NEF_ROW
            if (h < EP - 1) h2p_barrier();      // one epilogue piece per tick
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
        h2p_barrier();      // the last epilogue tick
    }
    for (int i = 0; i < (NGR - 1 - group) * OFFT; ++i) h2p_barrier();      // group 0 idles while group 1 finishes
}
#undef a

#undef H2P_RELOAD_ARGS

template <int K, int PRO>
int launch_h2p(const nef_conv_args& a, hipStream_t st) {
    constexpr size_t lds = (size_t)NGR * GRP_LDS;
    static unsigned long long lds_set = 0;
    if (int e = nef_ensure_dyn_lds(reinterpret_cast<const void*>(&conv_h2p_kernel<K, PRO>), lds, &lds_set)) return e;
    const int tps = (a.T + NTO - 1) / NTO;
    const int nb8 = (a.B + 7) / 8;
    const int m_tiles = a.Cout_g / MT;
    const int64_t total = (int64_t)a.G * m_tiles * nb8 * 8 * tps;      // work ids, incl. the empty tiles of samples B .. 8 nb8 - 1
    if (total <= 0 || total > 0x3fffffff) return NEF_E_SHAPE;
    // one resident workgroup of twelve waves per CU (a 6-wave workgroup does not share a CU with a second one at 168 registers: the
    // dispatcher wants 2 + 2 + 1 + 1 wave slots on the four SIMDs twice); a multiple of 8, so that a stream keeps its id mod 8
    const int per_cu = nef_opt_get(NEF_OPT_H2P_WGS);
    int64_t grid = (int64_t)(per_cu > 0 ? per_cu : 1) * nef_cu_count();
    grid -= grid % 8;
    if (grid < 8) grid = 8;
    if (NGR * grid > total) grid = (total / NGR + 7) / 8 * 8;
    const int my_max = (int)((total + NGR * grid - 1) / (NGR * grid));
    hipLaunchKernelGGL((conv_h2p_kernel<K, PRO>), dim3((unsigned)grid), dim3(NTHREADS), lds, st, a, tps, nb8, m_tiles, my_max);
    return nef_launch_status();
}

}  // namespace

extern "C" {
// diagnostics (not in the header): resident workgroups per CU the runtime reports for conv_h2p_kernel<3, 0>
int nef_debug_h2p_occupancy(void) {
    int n = -1;
    constexpr size_t lds = (size_t)NGR * GRP_LDS;
    static unsigned long long lds_set = 0;
    nef_ensure_dyn_lds(reinterpret_cast<const void*>(&conv_h2p_kernel<3, 0>), lds, &lds_set);
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&conv_h2p_kernel<3, 0>), NTHREADS, lds);
    return e == hipSuccess ? n : -(int)e;
}
}

// shapes the producer / consumer form takes (the caller has checked nef_h2_ok): whole 256-column tiles of one sample (T >= 128),
// at least three 16-channel stages (the producers' loads run two stages ahead of their stores)
__attribute__((visibility("hidden"))) bool nef_h2p_ok(const nef_conv_args* a) {
    const int Tin = (a->pro_mode & 2) ? a->T / 2 : a->T;
    return a->T >= NTO / 2 && a->Cin_g / KC >= 3 && a->Cout_g % MT == 0 && (int64_t)8 * Tin * 4 < 0x7fffffff &&
           (!(a->pro_mode & 2) || a->T % 4 == 0);
}

__attribute__((visibility("hidden"))) int nef_h2p_launch(const nef_conv_args* a, hipStream_t st) {
    if (!nef_h2p_ok(a)) return NEF_E_SHAPE;
    if ((a->pro_mode & 1) && !(a->pro_a && a->pro_b && a->pro_Bp > 0)) return NEF_E_NULL;
    if (a->K == 7) return launch_h2p<7, 0>(*a, st);
    if (a->K == 1) return launch_h2p<1, 0>(*a, st);
    switch (a->pro_mode) {
        case 0: return launch_h2p<3, 0>(*a, st);
        case 1: return launch_h2p<3, 1>(*a, st);
        case 2: return launch_h2p<3, 2>(*a, st);
        default: return launch_h2p<3, 3>(*a, st);
    }
}
