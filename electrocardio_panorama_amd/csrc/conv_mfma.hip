// Grouped Conv1d (stride 1, K in {1,3,7}) forward / backward-data / backward-weight as implicit GEMMs on
// the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, fmaf-chain numerics).
//
// Replaces nn.Conv1d at reference codes/network/model_nefnet.py:18,21,32,44 and
// codes/network/encoder/resnet_1d.py:23 (forward), and the autograd-derived convolution_backward.
//
// GEMM view, per group g (all time-contiguous, so the N / reduction axis is the coalesced one):
//   forward / bwd-data : Y[co][n] = sum_{ci,k} Wp[k][ci][co] * X[ci][n + k - pad]     M=co  N=(b,t)  K=(ci,k)
//   bwd-weight         : gW[co][ci][k] = sum_{n} gY[co][n] * X[ci][n + k - pad]       M=co  N=ci     K=(b,t)
// A column tile never straddles a sample: long sequences are cut into 128(64)-column tiles, short ones
// (the fixed 16/32-sample ROI latents) pack several whole samples per tile, each with its own zero halo in LDS.
#include "nef_common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Build-time split (round 6): this file is compiled FOUR times, -DNEF_MFMA_PART=1..4, into four objects that build in parallel
// (one translation unit took 52 s -- the long pole of the library build).  Every part sees all templates; a part only INSTANTIATES
// what its own dispatch code names: 1 = the C entry points + the direct kernels (forward, weight gradient), packs, channel sums;
// 2 = the F(4,.) forward / backward-data kernels; 3 = the F(2,.) ones; 4 = the transposed-Winograd weight gradients.  Parts 2..4
// are reached from part 1 through the hidden nef_mfma_* functions below.  NEF_MFMA_PART undefined / 0: everything in one object.
#ifndef NEF_MFMA_PART
#define NEF_MFMA_PART 0
#endif
#define NEF_PART(n) (NEF_MFMA_PART == 0 || NEF_MFMA_PART == (n))

namespace {

// NEF_ABL: timing-only ablation builds (tools/ablate_k7.py; results are WRONG by construction, never shipped): bit0 = no
// weight (A) fetches inside the main loop, bit1 = no activation fetches / LDS staging stores inside the loop, bit2 = no LDS
// fragment reads and no transform arithmetic inside the loop (operands are loop-invariant registers of random data),
// bit3 = no epilogue (the accumulators stay live behind a run-time-false branch).  15 = the MFMA stream and its barriers.
#ifndef NEF_ABL
#define NEF_ABL 0
#endif
// NEF_TRACE: measurement builds only (tools/trace_conv.py).  Wave 0 of every 16th workgroup of conv_wino4_kernel keeps
// s_memtime stamps of its phases (entry, first tile staged, every stage's barrier, epilogue done) and writes them out at exit.
#ifdef NEF_TRACE
__device__ unsigned long long* nef_trace_ptr = nullptr;
#define NEF_TR(I) if (tr_on) tr_t[(I)] = __builtin_readcyclecounter();
#else
#define NEF_TR(I)
#endif
constexpr int NT = 128;   // forward: columns per workgroup
constexpr int WT = 64;    // bwd-weight: reduction columns per staged tile

template <int K> struct StageK;
template <> struct StageK<7> { static constexpr int KC = 16; };
// Forward stage size per (K, TM).  The 128-row K=7 kernel stages 8 channels at a time: 34 KB of LDS and 162 VGPRs put 3
// workgroups on a CU instead of 2, so one more workgroup's MFMA stream covers the others' barriers and epilogues:
// 126.7 -> 133.1 TFLOP/s on the encoder convs (80.5 -> 84.6 % of the fp32 matrix peak).  The 64-row variants cannot
// split an 8-channel weight tile evenly over 256 threads and keep 16.  The 128-row K=3 kernels take 8 as well (-0.4 ms).
#ifndef NEF_K7_KC
#define NEF_K7_KC 8
#endif
#ifndef NEF_K3_KC2
#define NEF_K3_KC2 8
#endif
template <int K, int TM> struct FwdStage {
    static constexpr int KC = (K == 7 && TM == 2) ? NEF_K7_KC : (K == 3 && TM == 2) ? NEF_K3_KC2 : StageK<K>::KC;
};
// K=3: 16-channel stages keep the kernel at <= 168 VGPRs and 34 KB of LDS -> 3 workgroups per CU (the per-tile fixed
// costs of these short-K convs then overlap across workgroups): -2.3 % step time against 32-channel stages.
template <> struct StageK<3> { static constexpr int KC = 16; };
template <> struct StageK<1> { static constexpr int KC = 64; };

struct ColTiling {
    int seg_shift;   // log2(columns per sample segment inside a tile)
    int nseg;        // samples per tile
    int tps;         // tiles per sample (nseg == 1) else 1
    int n_tiles;
};

static ColTiling make_tiling(int B, int T, int tile_cols) {
    ColTiling c;
    if (T >= tile_cols) {
        c.seg_shift = 31 - __builtin_clz(tile_cols);
        c.nseg = 1;
        c.tps = (T + tile_cols - 1) / tile_cols;
        c.n_tiles = B * c.tps;
    } else {
        int seg = 16;
        while (seg < T) seg <<= 1;
        c.seg_shift = 31 - __builtin_clz(seg);
        c.nseg = tile_cols / seg;
        c.tps = 1;
        c.n_tiles = (B + c.nseg - 1) / c.nseg;
    }
    return c;
}

// ------------------------------------------------------------------------------------------------
// forward / bwd-data
// ------------------------------------------------------------------------------------------------
// PRO: input prologue applied while staging (decoder fusion): bit0 = BatchNorm affine + ReLU of the producing layer,
// bit1 = the input is stored at half resolution and is x2-upsampled (linear, align_corners=False) on the fly.
template <int K, int TM, int PRO>
__global__ __launch_bounds__(256, 2) void conv_fwd_kernel(nef_conv_args a, int seg_shift, int nseg, int tps,
                                                         int n_tiles, int m_tiles) {
    constexpr bool UP = (PRO & 2) != 0, AFF = (PRO & 1) != 0;
    constexpr int NS = UP ? 2 : 1;
    constexpr int KC = FwdStage<K, TM>::KC;
    constexpr int MT = 64 * TM;
    constexpr int PAD = (K - 1) / 2;
    constexpr int XRS = NT + (NT / 16) * (K - 1);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                  // [K][KC][MT]
    float* Xl = smem + K * KC * MT;    // [KC][XRS]

    const int tile = blockIdx.x % n_tiles;
    const int gm = blockIdx.x / n_tiles;
    const int mt = gm % m_tiles;
    const int g = gm / m_tiles;
    const int seg = 1 << seg_shift;
    int b0, t0;
    if (nseg == 1) {
        // Column tiles of one sample share their boundary cache lines (rows are not 128-byte multiples) and their halo
        // columns.  Workgroup ids go round-robin over the 8 XCDs, so the tiles of a sample are issued 8 ids apart: same
        // XCD, back to back -- the shared lines are then L2 hits instead of second fetches.
        const int full = (n_tiles / (8 * tps)) * (8 * tps);
        if (tile < full) {
            const int grp = tile / (8 * tps), r = tile % (8 * tps);
            b0 = grp * 8 + (r & 7);
            t0 = (r >> 3) * NT;
        } else {
            b0 = tile / tps;
            t0 = (tile - b0 * tps) * NT;
        }
    } else {
        b0 = tile * nseg;
        t0 = 0;
    }
    const int m0 = mt * MT;
    const int T = a.T, Cig = a.Cin_g, Cog = a.Cout_g;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int segw = seg + K - 1;
    const int xrow = nseg * segw;

    // Staging through buffer descriptors (nef_common.h): per lane ONE byte offset per row position, rows are
    // selected by a wave-uniform SGPR offset, out-of-range positions (halo beyond the sample, batch tail) carry
    // NEF_OOB and read as 0.0 -- so a stage is one burst of independent loads with no branches.
    constexpr int NIT = (K == 1) ? 2 : 3;          // xrow <= 128 (K=1) or <= 176
    constexpr int XR = KC / 4;                     // activation rows per wave
    constexpr int M4 = MT / 4;
    constexpr int NW = K * KC * M4 / 256;          // float4 weight loads per thread and stage
    constexpr int RQ = 256 / M4;                   // weight rows covered per pass of the workgroup
    static_assert(K * KC * M4 % 256 == 0, "weight tile must split evenly over the workgroup");
    static_assert(KC % RQ == 0, "rows per pass must divide the channel chunk");
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t xrs = nef_rsrc(a.x + (int64_t)b0 * a.x_bs + (int64_t)g * a.x_gs);
    const __amdgpu_buffer_rsrc_t wrs = nef_rsrc(a.wp + (int64_t)g * K * Cig * Cog + m0);
    const int Tin = UP ? (T >> 1) : T;             // stored row length of the input
    unsigned xvo[NIT][NS];
    float lam[NIT];                                // weight of the second tap when upsampling
    int64_t soff[NIT];
    bool xok[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int r = lane + 64 * it;
        const int s = r / segw;
        const int u = r - s * segw;
        const int t = t0 + u - PAD;
        xok[it] = (r < xrow) && (b0 + s < a.B) && (t >= 0) && (t < T);
        lam[it] = 0.f;
        if constexpr (UP) {
            // nn.Upsample(scale_factor=2, 'linear', align_corners=False): src = 0.5*(t+0.5)-0.5, clamped at 0
            float src = 0.5f * ((float)t + 0.5f) - 0.5f;
            if (src < 0.f) src = 0.f;
            int i0 = (int)src;
            if (i0 > Tin - 1) i0 = Tin - 1;
            const int i1 = i0 + (i0 < Tin - 1 ? 1 : 0);
            lam[it] = src - (float)i0;
            xvo[it][0] = xok[it] ? (unsigned)(((int64_t)s * a.x_bs + i0) * 4) : NEF_OOB;
            xvo[it][NS - 1] = xok[it] ? (unsigned)(((int64_t)s * a.x_bs + i1) * 4) : NEF_OOB;
        } else {
            xvo[it][0] = xok[it] ? (unsigned)(((int64_t)s * a.x_bs + t) * 4) : NEF_OOB;
        }
        soff[it] = xok[it] ? (int64_t)(b0 + s) * a.sc_bs + (int64_t)g * a.sc_gs : 0;
    }
    const int pro_row0 = AFF ? (b0 / a.pro_Bp) * a.G * Cig + g * Cig : 0;   // [pass][channel] base of pro_a / pro_b
    // weight tile: float4 index i = threadIdx.x + 256*q -> row rc = i / M4 = (kk, ci), column m4 = i % M4;
    // (kk, ci) of load q is a compile-time offset from this thread's first row.
    const unsigned wvo = (unsigned)(((int)(threadIdx.x / M4) * Cog + 4 * (int)(threadIdx.x % M4)) * 4);
    const int w_kstride = Cig * Cog;

    // LDS column offsets of this lane's two 32-column MFMA tiles
    int coloff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = wn * 64 + j * 32 + lo;
        coloff[j] = (col >> seg_shift) * segw + (col & (seg - 1));
    }

    f32x16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Register-staged pipeline: the loads of stage s+1 are issued before the MFMA loop of stage s and only
    // written to LDS after it, so HBM/L2 latency hides under the matrix work (one LDS buffer, two barriers).
    f32x4 wreg[NW];
    float xreg[XR][NIT][NS];
#define NEF_ISSUE_LOADS(C0)                                                                                          \
    {                                                                                                               \
        _Pragma("unroll") for (int q = 0; q < NW; ++q) wreg[q] = nef_buf_f32x4(                                     \
            wrs, wvo, (unsigned)((((q * RQ) / KC) * w_kstride + ((q * RQ) % KC + (C0)) * Cog) * 4));                 \
        _Pragma("unroll") for (int rr = 0; rr < XR; ++rr) {                                                         \
            const unsigned so = (unsigned)(((C0) + wave_u + 4 * rr) * Tin * 4);                                     \
            _Pragma("unroll") for (int it = 0; it < NIT; ++it)                                                      \
                _Pragma("unroll") for (int ns = 0; ns < NS; ++ns) xreg[rr][it][ns] = nef_buf_f32(xrs, xvo[it][ns], so); \
        }                                                                                                           \
    }
    NEF_ISSUE_LOADS(0)
    for (int c0 = 0; c0 < Cig; c0 += KC) {
        if (a.in_scale) {
#pragma unroll
            for (int rr = 0; rr < XR; ++rr)
#pragma unroll
                for (int it = 0; it < NIT; ++it)
                    xreg[rr][it][0] *= a.in_scale[soff[it] + (xok[it] ? c0 + wave + 4 * rr : 0)];
        }
        __syncthreads();                 // every wave is done reading the previous stage
        {
            f32x4* Wl4 = reinterpret_cast<f32x4*>(Wl);
#pragma unroll
            for (int q = 0; q < NW; ++q) Wl4[threadIdx.x + 256 * q] = wreg[q];
#pragma unroll
            for (int rr = 0; rr < XR; ++rr) {
                float pa = 1.f, pb = 0.f;
                if constexpr (AFF) {
                    pa = a.pro_a[pro_row0 + c0 + wave_u + 4 * rr];
                    pb = a.pro_b[pro_row0 + c0 + wave_u + 4 * rr];
                }
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int r = lane + 64 * it;
                    float v = xreg[rr][it][0];
                    if constexpr (AFF) v = fmaxf(fmaf(v, pa, pb), 0.f);
                    if constexpr (UP) {
                        float v1 = xreg[rr][it][NS - 1];
                        if constexpr (AFF) v1 = fmaxf(fmaf(v1, pa, pb), 0.f);
                        v = (1.f - lam[it]) * v + lam[it] * v1;
                    }
                    if constexpr (PRO != 0) v = xok[it] ? v : 0.f;      // padding is applied AFTER the prologue
                    if (r < xrow) Xl[(wave + 4 * rr) * XRS + r] = v;
                }
            }
        }
        __syncthreads();
        if (c0 + KC < Cig) NEF_ISSUE_LOADS(c0 + KC)
        // MFMA loop, software-pipelined in registers: the LDS fragments of k-step group gi+1 are read while the
        // MFMAs of group gi issue (fully unrolled, so every register index is static), one ds_read per MFMA slot.
        {
#ifndef NEF_GS
#define NEF_GS 2
#endif
            constexpr int GS = NEF_GS;                  // k-steps (of 2 channels) per group
            constexpr int SPK = KC / 2;                 // k-steps per tap
            constexpr int NG = K * SPK / GS;
            static_assert((K * SPK) % GS == 0, "k-steps must split into whole groups");
            float fa[2][GS][TM], fb[2][GS][2];
#define NEF_LOAD_GROUP(GI, BUF)                                                                                      \
    _Pragma("unroll") for (int s_ = 0; s_ < GS; ++s_) {                                                             \
        const int step_ = (GI) * GS + s_;                                                                           \
        const int kk_ = step_ / SPK, c_ = (step_ % SPK) * 2;                                                        \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                              \
            fa[BUF][s_][i] = Wl[(kk_ * KC + c_ + hi) * MT + (wm * TM + i) * 32 + lo];                               \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) fb[BUF][s_][j] = Xl[(c_ + hi) * XRS + coloff[j] + kk_];       \
    }
            NEF_LOAD_GROUP(0, 0)
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                if (gi + 1 < NG) NEF_LOAD_GROUP(gi + 1, (gi + 1) & 1)
#pragma unroll
                for (int s_ = 0; s_ < GS; ++s_)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[gi & 1][s_][i], fb[gi & 1][s_][j],
                                                                             acc[i][j], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < GS * (TM + 2); ++q) {      // interleave: one MFMA, one LDS read
                    __builtin_amdgcn_sched_group_barrier(0x008, (2 * TM * GS) / (GS * (TM + 2)) > 0 ? 1 : 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
#undef NEF_LOAD_GROUP
        }
    }

    // epilogue: bias, residual, ReLU, dropout, gate.  Per 32x32 tile the optional operands are fetched as 16
    // independent loads per lane before use; 32 lanes store 128 contiguous bytes per output row.
    const int64_t ctot = (int64_t)a.G * Cog;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = wn * 64 + j * 32 + lo;
        const int s = col >> seg_shift;
        const int tt = col & (seg - 1);
        const int b = b0 + s;
        const int t = t0 + tt;
        const bool live = (b < a.B) && (t < T);
        const int bs = live ? b : 0, ts = live ? t : 0;      // safe coordinates for the loads of dead lanes
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int cobase = m0 + (wm * TM + i) * 32 + 4 * hi;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r];
            if (a.bias) {
                float t16[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) t16[r] = a.bias[g * Cog + cobase + (r & 3) + 8 * (r >> 2)];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += t16[r];
            }
            if (a.res) {
                const float* rp = a.res + (int64_t)bs * a.res_bs + (int64_t)g * a.res_gs + (int64_t)cobase * T + ts;
                float t16[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) t16[r] = rp[(int64_t)((r & 3) + 8 * (r >> 2)) * T];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += t16[r];
            }
            if (a.relu) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (a.mask) {
                const uint8_t* mp = a.mask + ((int64_t)bs * ctot + (int64_t)g * Cog + cobase) * T + ts;
                float t16[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) t16[r] = (float)mp[(int64_t)((r & 3) + 8 * (r >> 2)) * T];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] *= t16[r] * a.drop_scale;
            } else if (a.drop_p > 0.f) {
                const int64_t d0 = ((int64_t)bs * ctot + (int64_t)g * Cog + cobase) * T + ts;
                const uint64_t seed = a.rng_seed + (a.rng_seed_dev ? a.rng_seed_dev[0] : 0ull);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint64_t dense = (uint64_t)(d0 + (int64_t)((r & 3) + 8 * (r >> 2)) * T);
                    v[r] = (nef_rng_uniform(seed, dense) >= a.drop_p) ? v[r] * a.drop_scale : 0.f;
                }
            }
            if (a.gate) {
                const float* gp = a.gate + (int64_t)bs * a.gate_bs + (int64_t)g * a.gate_gs + (int64_t)cobase * T + ts;
                float t16[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) t16[r] = gp[(int64_t)((r & 3) + 8 * (r >> 2)) * T];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = t16[r] > 0.f ? v[r] * a.gate_scale : 0.f;
            }
            if (live) {
                float* yp = a.y + (int64_t)b * a.y_bs + (int64_t)g * a.y_gs + (int64_t)cobase * T + t;
#pragma unroll
                for (int r = 0; r < 16; ++r) yp[(int64_t)((r & 3) + 8 * (r >> 2)) * T] = v[r];
            }
        }
    }
}

#undef NEF_ISSUE_LOADS

template <int K, int TM, int PRO = 0>
static int launch_conv_fwd(const nef_conv_args& a, hipStream_t st) {
    constexpr int KC = FwdStage<K, TM>::KC;
    constexpr int MT = 64 * TM;
    constexpr int XRS = NT + (NT / 16) * (K - 1);
    constexpr size_t lds = (size_t)(K * KC * MT + KC * XRS) * sizeof(float);
    static unsigned long long lds_set = 0;      // per-device bits, see nef_ensure_dyn_lds
    if (int e = nef_ensure_dyn_lds(reinterpret_cast<const void*>(&conv_fwd_kernel<K, TM, PRO>), lds, &lds_set)) return e;
    const ColTiling ct = make_tiling(a.B, a.T, NT);
    const int m_tiles = a.Cout_g / MT;
    const int64_t blocks = (int64_t)a.G * m_tiles * ct.n_tiles;
    if (blocks <= 0 || blocks > 0x7fffffff) return NEF_E_SHAPE;
    hipLaunchKernelGGL((conv_fwd_kernel<K, TM, PRO>), dim3((unsigned)blocks), dim3(256), lds, st, a, ct.seg_shift, ct.nseg,
                       ct.tps, ct.n_tiles, m_tiles);
    return nef_launch_status();
}

// ------------------------------------------------------------------------------------------------
// K = 3 through Winograd F(2,3): 4 multiplies per 2 outputs instead of 6 -- 2/3 of the matrix-core work
// ------------------------------------------------------------------------------------------------
// For output pair j of a row (outputs 2j, 2j+1; inputs d_m = x[2j-1+m], m = 0..3, zero padded) and taps g0..g2:
//     v0 = d0-d2   v1 = d1+d2   v2 = d2-d1   v3 = d1-d3              (input transform)
//     u0 = g0   u1 = (g0+g1+g2)/2   u2 = (g0-g1+g2)/2   u3 = g2      (weight transform, nef_pack_weight_wino)
//     M_i[co][j] = sum_ci u_i[ci][co] * v_i[ci][j]                   (4 GEMMs over ci on the matrix cores)
//     y[2j] = M0+M1+M2   y[2j+1] = M1-M2-M3                          (output transform, in the epilogue)
// The activation tile is staged RAW exactly as in conv_fwd_kernel (so in_scale and both input prologues work
// unchanged); the input transform happens on the way from LDS to the MFMA B operand: one lane owns pair j, reads
// x[2j..2j+3] of a channel row as two conflict-free ds_read_b64 and forms v0..v3 with four VALU ops.  A wave owns
// 64 output channels x 32 pairs (64 outputs) = 4 x 2 accumulator tiles; a workgroup is WM x (4/WM) waves:
// WM = 2 -> 128 channels x 128 outputs, WM = 1 -> 64 channels x 256 outputs.  Still exact fp32 multiplies and adds;
// the result differs from the direct form by the rounding of the three transforms (transform entries are 0, +-1,
// 1/2: a few ulp, measured in tests/test_ops_gpu.py::test_conv_winograd).  Sequences shorter than a tile keep the
// direct kernel.  NEF_WINOGRAD=0 in the environment disables this path (ops.py).
#ifndef NEF_WKC
#define NEF_WKC 16
#endif
constexpr int WKC = NEF_WKC;   // channels per activation stage: 64 MFMAs per wave between barriers
constexpr int PRO_MAX_CIN = 512; // input channels per group the LDS table of the affine prologue holds

// Operand paths.  B (activations): raw tile through LDS, DOUBLE-buffered -- the registers holding stage s+1 (fetched
// during the MFMA loop of stage s) are written to the other buffer right after that loop, so there is ONE barrier per
// stage and nothing between it and the next MFMA.  A (transformed weights, [g][i][ci][co]): never touches LDS -- a lane's
// A fragment for (i, channel pair, co tile) is 32 consecutive floats of one packed row, i.e. a perfectly coalesced
// 128-byte line per half-wave, so every wave fetches its own fragments straight from L2/L1 (the whole operand is
// <= 0.5 MB per group and stays resident) three k-steps ahead of use into four rotating register sets.  That takes
// the weight tile (32 KB per stage), its LDS writes and 8 of the 10 LDS reads per k-step out of the kernel.
//
// K = 7 (round 3) uses the same machinery on the taps split 4 + 3: F(2,4) -- points 0, 1, -1, 2, inf: 5 products for 2
// outputs of a 4-tap filter -- takes taps 0..3 on x[2j-3 .. 2j+1], F(2,3) takes taps 4..6 on x[2j+1 .. 2j+4]:
//     v = B^T d:  v0 = 2d0-d1-2d2+d3   v1 = -2d1-d2+d3   v2 = 2d1-3d2+d3   v3 = -d1+d3   v4 = 2d1-d2-2d3+d4
//     u = G g:    u0 = g0/2   u1 = -(g0+g1+g2+g3)/2   u2 = (-g0+g1-g2+g3)/6   u3 = (g0+2g1+4g2+8g3)/6   u4 = g3
//     y[2j] = M0+M1+M2+M3      y[2j+1] = M1-M2+2*M3+M4
// The F(2,3) group's products have the output columns (1,0), (1,1), (1,-1), (0,-1): the first three accumulate into M0, M1, M2
// and the fourth, with its plane negated by the pack, into M4 -- five M tiles per co tile, 5 + 4 = 9 multiplies per channel
// pair (rounds 1-2: 3 + 3 + 1 through two F(2,3) groups = 10; direct: 14).  Like F(2,3), F(2,4) keeps the reference's exact
// zeros: every product only sees inputs inside the receptive field of the outputs it feeds (M0: d0..d3 -> y0 only; M1..M3:
// d1..d3; M4: d1..d4 -> y1 only), so an output over the all-zero tail of a beat is a sum of exact zeros.  Entries up to 3
// and 4/3: a little more rounding than F(2,3), far less than F(4,.); the 3-step SGD trajectory stays inside its 2e-4 bar.
// A lane reads x[2j-3 .. 2j+4] of a channel row as four aligned ds_read_b64; 9 operand planes (+ one of padding in the
// 16-byte layout); 160 accumulator VGPRs, 246 in all, no spills.
template <int K, int WM, int PRO>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(nef_conv_args a, int tps, int n_tiles, int m_tiles) {
    constexpr bool UP = (PRO & 2) != 0, AFF = (PRO & 1) != 0;
    constexpr int NS = UP ? 2 : 1;
    constexpr int KC = WKC;
    constexpr int WN = 4 / WM;
    constexpr int PAD = (K - 1) / 2;
    constexpr int NPL = K == 3 ? 4 : 10; // weight planes per (ci, co)
    constexpr int MT = 64 * WM;          // output channels per workgroup
    constexpr int NTO = 64 * WN;         // outputs (columns) per workgroup
    constexpr int XROW = NTO + K - 1;    // staged positions per channel row: t0-PAD .. t0+NTO+PAD-1
    constexpr int XRS = NTO + 16;        // LDS row pitch (even: rows stay 8-byte aligned for ds_read_b64)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xl = smem;                    // [2][KC][XRS]
    float* Pl = smem + 2 * KC * XRS;     // [2][Cin_g]: prologue affine of this tile's pass (AFF only)

    const int tile = blockIdx.x % n_tiles;
    const int gm = blockIdx.x / n_tiles;
    const int mt = gm % m_tiles;
    const int g = gm / m_tiles;
    int b0, t0;
    {   // tiles of one sample 8 workgroup ids apart: same XCD, shared boundary lines hit in that L2 (see conv_fwd_kernel)
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
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    constexpr int NIT = (XROW + 63) / 64;
    constexpr int XR = KC / 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm_u = wave_u / WN;
    const float* const xbase = a.x + (int64_t)b0 * a.x_bs + (int64_t)g * a.x_gs;
    const __amdgpu_buffer_rsrc_t xrs = nef_rsrc(xbase);
    // K = 7: q slabs of 16-byte vectors; K = 3: planes [ci][co] (see the A-operand note below)
    const int a_rstride = K == 7 ? (Cog >> 6) * 128 : Cog;   // floats per reduction channel inside one slab / plane
    const int a_qstride = Cig * a_rstride;                   // floats per slab / plane
    const __amdgpu_buffer_rsrc_t wrs = nef_rsrc(a.wp + (int64_t)g * NPL * Cig * Cog +
                                                (K == 7 ? ((m0 >> 6) + wm_u) * 128 : m0 + wm_u * 64));
    const int Tin = UP ? (T >> 1) : T;
    unsigned xvo[NIT][NS];
    float lam[NIT];
    bool xok[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int r = lane + 64 * it;
        const int t = t0 + r - PAD;
        xok[it] = (r < XROW) && (t >= 0) && (t < T);
        lam[it] = 0.f;
        if constexpr (UP) {
            float src = 0.5f * ((float)t + 0.5f) - 0.5f;
            if (src < 0.f) src = 0.f;
            int i0 = (int)src;
            if (i0 > Tin - 1) i0 = Tin - 1;
            const int i1 = i0 + (i0 < Tin - 1 ? 1 : 0);
            lam[it] = src - (float)i0;
            xvo[it][0] = xok[it] ? (unsigned)(i0 * 4) : NEF_OOB;
            xvo[it][NS - 1] = xok[it] ? (unsigned)(i1 * 4) : NEF_OOB;
        } else {
            xvo[it][0] = xok[it] ? (unsigned)(t * 4) : NEF_OOB;
        }
    }
    const int64_t soff = (int64_t)b0 * a.sc_bs + (int64_t)g * a.sc_gs;
    const int pro_row0 = AFF ? (b0 / a.pro_Bp) * a.G * Cig + g * Cig : 0;
    const unsigned avo = (unsigned)((hi * a_rstride + (K == 7 ? 4 : 1) * lo) * 4);     // reduction channel (2*step + hi), lane lo

    constexpr int NACC = K == 7 ? 5 : 4;      // M tiles per co tile (K = 7: the point-2 tile of the F(2,4) group)
    f32x16 acc[NACC][2];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][tm][r] = 0.f;

    constexpr int SPK = KC / 2;          // k-steps (channel pairs) per stage
    // A fragments are fetched AHEAD k-steps ahead into NSET rotating register sets (a K = 7 step is 20 MFMAs long, one
    // step of lead covers the L2 latency; the 8-MFMA steps of K = 3 need three)
    constexpr int AHEAD = K == 3 ? 3 : 1;
    constexpr int NSET = AHEAD + 1;
    static_assert(SPK % NSET == 0, "the A sets must line up across stages");
    const int nsteps = Cig / 2;
    // A operand: nef_pack_weight_wino lays the 2*NPL values a lane needs per k-step (plane i, co tile tm) out as NQ 16-byte
    // vectors, [g][q][ci][64-wide co block][lo][4] with value 2*i + tm = 4*q + e: one buffer_load_dwordx4 per four MFMAs
    // (5 instead of 20 vector-memory instructions per K = 7 k-step; 512 contiguous bytes per half-wave).  q is the
    // OUTER index on purpose: the NQ loads of a k-step then go to addresses >= 64 KB apart, i.e. to different L2
    // channels -- with q innermost (one contiguous 5 KB per k-step, which every resident workgroup requests at about
    // the same time) the gain of the wide loads was half as large.
    // K = 3 (8 values per k-step, fetched 3 steps ahead) keeps the plane-major operand [g][plane][ci][co] and dword loads:
    // measured 2..4 % FASTER than two 16-byte loads per step on every F(2,3) K = 3 shape, while K = 7 gains 2 % from them.
    constexpr bool AV4 = K == 7;
    constexpr int NQ = NPL / 2;
    f32x4 fa4[AV4 ? NSET : 1][NQ];
    float fa1[AV4 ? 1 : NSET][NPL][2];
#define NEF_FA(SET, I, TM) (AV4 ? fa4[AV4 ? (SET) : 0][(2 * (I) + (TM)) >> 2][(2 * (I) + (TM)) & 3] : fa1[AV4 ? 0 : (SET)][I][TM])
#define NEF_WA_ISSUE(GS, SET)                                                                                        \
    {                                                                                                               \
        const int gs_ = (GS) < nsteps ? (GS) : nsteps - 1;      /* past the end: a harmless repeat of the last step */ \
        if constexpr (AV4) {                                                                                        \
            _Pragma("unroll") for (int q = 0; q < NQ; ++q)                                                          \
                fa4[AV4 ? (SET) : 0][q] = nef_buf_f32x4(wrs, avo, (unsigned)((q * a_qstride + 2 * gs_ * a_rstride) * 4)); \
        } else {                                                                                                    \
            _Pragma("unroll") for (int i = 0; i < NPL; ++i)                                                         \
                _Pragma("unroll") for (int tm = 0; tm < 2; ++tm)                                                    \
                    fa1[AV4 ? 0 : (SET)][i][tm] =                                                                   \
                        nef_buf_f32(wrs, avo, (unsigned)((i * a_qstride + 2 * gs_ * a_rstride + tm * 32) * 4));     \
        }                                                                                                           \
    }
    float xreg[XR][NIT][NS];
#define NEF_WX_ISSUE(C0, RS)                                                                                         \
    {                                                                                                               \
        _Pragma("unroll") for (int rr = 0; rr < XR; ++rr) {                                                         \
            const unsigned so = (unsigned)(((C0) + wave_u + 4 * rr) * Tin * 4);                                     \
            _Pragma("unroll") for (int it = 0; it < NIT; ++it)                                                      \
                _Pragma("unroll") for (int ns = 0; ns < NS; ++ns) xreg[rr][it][ns] = nef_buf_f32(RS, xvo[it][ns], so); \
        }                                                                                                           \
    }
#define NEF_WX_STORE(C0, BUFP)                                                                                       \
    {                                                                                                               \
        if (a.in_scale) {                                                                                           \
            _Pragma("unroll") for (int rr = 0; rr < XR; ++rr) {                                                     \
                const float sc = a.in_scale[soff + (C0) + wave + 4 * rr];                                           \
                _Pragma("unroll") for (int it = 0; it < NIT; ++it) xreg[rr][it][0] *= sc;                           \
            }                                                                                                       \
        }                                                                                                           \
        _Pragma("unroll") for (int rr = 0; rr < XR; ++rr) {                                                         \
            float pa = 1.f, pb = 0.f;                                                                               \
            if constexpr (AFF) {                                                                                    \
                pa = Pl[(C0) + wave_u + 4 * rr];                                                                    \
                pb = Pl[Cig + (C0) + wave_u + 4 * rr];                                                              \
            }                                                                                                       \
            _Pragma("unroll") for (int it = 0; it < NIT; ++it) {                                                    \
                const int r = lane + 64 * it;                                                                       \
                float v = xreg[rr][it][0];                                                                          \
                if constexpr (AFF) v = fmaxf(fmaf(v, pa, pb), 0.f);                                                 \
                if constexpr (UP) {                                                                                 \
                    float v1 = xreg[rr][it][NS - 1];                                                                \
                    if constexpr (AFF) v1 = fmaxf(fmaf(v1, pa, pb), 0.f);                                           \
                    v = (1.f - lam[it]) * v + lam[it] * v1;                                                         \
                }                                                                                                   \
                if constexpr (PRO != 0) v = xok[it] ? v : 0.f;                                                      \
                if (r < XROW) (BUFP)[(wave + 4 * rr) * XRS + r] = v;                                                \
            }                                                                                                       \
        }                                                                                                           \
    }
    NEF_WX_ISSUE(0, xrs)
#pragma unroll
    for (int s_ = 0; s_ < ((NEF_ABL & 1) ? NSET : AHEAD); ++s_) NEF_WA_ISSUE(s_, s_)
    if constexpr (AFF) {     // the producing BatchNorm's (a, b) of this tile's pass: one table in LDS instead of a global load per
                             // staged row (each of those was waited for with the queue drained)
        for (int i = threadIdx.x; i < Cig; i += 256) {
            Pl[i] = a.pro_a[pro_row0 + i];
            Pl[Cig + i] = a.pro_b[pro_row0 + i];
        }
        __syncthreads();
    }
    NEF_WX_STORE(0, Xl)
    __syncthreads();
    int st = 0;
    constexpr int NXV = K == 3 ? 2 : 4;        // ds_read_b64 per lane and k-step: x[2j-PAD .. 2j-PAD+2*NXV)
    f32x2 fx[2][NXV];
#define NEF_WX_LOAD(S, BUF)                                                                                          \
    {                                                                                                               \
        const f32x2* xp_ = reinterpret_cast<const f32x2*>(xb + 2 * (S) * XRS);                                      \
        _Pragma("unroll") for (int q_ = 0; q_ < NXV; ++q_) fx[BUF][q_] = xp_[q_];                                   \
    }
    if constexpr ((NEF_ABL & 4) != 0) {
        const float* xb = Xl + hi * XRS + 2 * (wn * 32 + lo);
        NEF_WX_LOAD(0, 0)
        NEF_WX_LOAD(1, 1)
    }
    for (int c0 = 0; c0 < Cig; c0 += KC, ++st) {
        const float* xb = Xl + ((NEF_ABL & 2) ? 0 : (st & 1)) * (KC * XRS) + hi * XRS + 2 * (wn * 32 + lo);
        const bool more = c0 + KC < Cig;
        const __amdgpu_buffer_rsrc_t xrs_n = nef_rsrc_n(xbase, more ? 0x7FFFFFFCu : 0u);
        if constexpr (!(NEF_ABL & 4)) NEF_WX_LOAD(0, 0)
#pragma unroll
        for (int s_ = 0; s_ < SPK; ++s_) {
            if constexpr (!(NEF_ABL & 1)) NEF_WA_ISSUE(st * SPK + s_ + AHEAD, (s_ + AHEAD) % NSET)
            // the activation rows of the next stage are requested once per stage, right behind an A request: the first
            // A fragment that is YOUNGER than them is consumed later in the stage, by when they have long arrived
            // (vector-memory results return in order)
            // (branch-free on purpose: past the last stage the burst goes through an empty descriptor.  With `if (more)`
            // around it the compiler's s_waitcnt bookkeeping took the smaller count of the two paths and made every wave sit
            // on the burst within the first k-step of the stage)
            if constexpr (!(NEF_ABL & 2)) if (s_ == 0) NEF_WX_ISSUE(c0 + KC, xrs_n)
            if constexpr (!(NEF_ABL & 4)) if (s_ + 1 < SPK) NEF_WX_LOAD(s_ + 1, (s_ + 1) & 1)
            const f32x2* d = fx[s_ & 1];
#define w_(I, TM) NEF_FA(s_ % NSET, I, TM)
            // One s_setprio per k-step.  Measured -3..5 % on the K = 3 shapes and -1 % on K = 7 (tools/bench_conv.py); a
            // single s_setprio(1) in front of the loop does nothing, so the gain is not the priority itself: the
            // instruction is a scheduling fence for the compiler and keeps each step's operand fetches, transforms and
            // MFMAs together instead of letting them drift across steps.
            __builtin_amdgcn_s_setprio(1);
            if constexpr (K == 3) {
                float v[4];
                v[0] = d[0][0] - d[1][0];
                v[1] = d[0][1] + d[1][0];
                v[2] = d[1][0] - d[0][1];
                v[3] = d[0][1] - d[1][1];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
                        acc[i][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(w_(i, tm), v[i], acc[i][tm], 0, 0, 0);
            } else {
                // taps 0..3 through F(2,4) on x[2j-3 .. 2j+1] (points 0, 1, -1, 2, inf): planes 0..4 into tiles 0..4
                const float x0 = d[0][0], x1 = d[0][1], x2 = d[1][0], x3 = d[1][1], x4 = d[2][0], x5 = d[2][1], x6 = d[3][0],
                            x7 = d[3][1];
                float v[5];
                const float p13 = x3 - x1;
                v[0] = fmaf(2.f, x0 - x2, p13);                  //  2x0 - x1 - 2x2 + x3
                v[1] = fmaf(-2.f, x1, x3 - x2);                  // -2x1 - x2 + x3
                v[2] = fmaf(2.f, x1, fmaf(-3.f, x2, x3));        //  2x1 - 3x2 + x3
                v[3] = p13;                                      //  -x1 + x3
                v[4] = fmaf(2.f, x1 - x3, x4 - x2);              //  2x1 - x2 - 2x3 + x4
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
                        acc[i][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(w_(i, tm), v[i], acc[i][tm], 0, 0, 0);
                // taps 4..6 through F(2,3) on x[2j+1 .. 2j+4]: its four products have the output columns (1,0), (1,1), (1,-1),
                // (0,-1) -- tiles 0, 1, 2 and, with the plane negated by the pack, tile 4
                float u[4];
                u[0] = x4 - x6;
                u[1] = x5 + x6;
                u[2] = x6 - x5;
                u[3] = x5 - x7;
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    acc[0][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(w_(5, tm), u[0], acc[0][tm], 0, 0, 0);
                    acc[1][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(w_(6, tm), u[1], acc[1][tm], 0, 0, 0);
                    acc[2][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(w_(7, tm), u[2], acc[2][tm], 0, 0, 0);
                    acc[4][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(w_(8, tm), u[3], acc[4][tm], 0, 0, 0);
                }
            }
        }
#undef w_
        if constexpr (!(NEF_ABL & 2)) if (more) NEF_WX_STORE(c0 + KC, Xl + ((st + 1) & 1) * (KC * XRS))
        __syncthreads();
    }
#undef NEF_WX_LOAD
#undef NEF_WA_ISSUE
#undef NEF_FA
#undef NEF_WX_ISSUE
#undef NEF_WX_STORE
    if constexpr ((NEF_ABL & 8) != 0) if (a.T >= 0) return;      // run-time true: the epilogue below is dead at run time only

    // epilogue: output transform, then bias / residual / ReLU / dropout / gate exactly as conv_fwd_kernel, on the two
    // adjacent outputs (2j, 2j+1) a lane owns per channel row: 8-byte loads and stores, 256 contiguous bytes per row.
    const int64_t ctot = (int64_t)a.G * Cog;
    const int t = t0 + 2 * (wn * 32 + lo);
    const bool live = (b0 < a.B) && (t < T);
    const int ts = live ? t : 0;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        const int cobase = m0 + (wm * 2 + tm) * 32 + 4 * hi;
        float y0[16], y1[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if constexpr (K == 3) {
                y0[r] = (acc[0][tm][r] + acc[1][tm][r]) + acc[2][tm][r];
                y1[r] = (acc[1][tm][r] - acc[2][tm][r]) - acc[3][tm][r];
            } else {      // y0 = M0+M1+M2+M3, y1 = M1-M2+2*M3+M4
                y0[r] = (acc[0][tm][r] + acc[1][tm][r]) + (acc[2][tm][r] + acc[3][tm][r]);
                y1[r] = (acc[1][tm][r] - acc[2][tm][r]) + fmaf(2.f, acc[3][tm][r], acc[NACC - 1][tm][r]);
            }
        }
        if (a.bias) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float bv = a.bias[g * Cog + cobase + (r & 3) + 8 * (r >> 2)];
                y0[r] += bv;
                y1[r] += bv;
            }
        }
        if (a.res) {
            const float* rp = a.res + (int64_t)b0 * a.res_bs + (int64_t)g * a.res_gs + (int64_t)cobase * T + ts;
            f32x2 t16[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                t16[r] = *reinterpret_cast<const f32x2*>(rp + (int64_t)((r & 3) + 8 * (r >> 2)) * T);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                y0[r] += t16[r][0];
                y1[r] += t16[r][1];
            }
        }
        if (a.relu) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                y0[r] = fmaxf(y0[r], 0.f);
                y1[r] = fmaxf(y1[r], 0.f);
            }
        }
        if (a.mask) {
            const uint8_t* mp = a.mask + ((int64_t)b0 * ctot + (int64_t)g * Cog + cobase) * T + ts;
            unsigned short t16[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                t16[r] = *reinterpret_cast<const unsigned short*>(mp + (int64_t)((r & 3) + 8 * (r >> 2)) * T);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                y0[r] *= (float)(t16[r] & 0xff) * a.drop_scale;
                y1[r] *= (float)(t16[r] >> 8) * a.drop_scale;
            }
        } else if (a.drop_p > 0.f) {
            const int64_t d0 = ((int64_t)b0 * ctot + (int64_t)g * Cog + cobase) * T + ts;
            const uint64_t seed = a.rng_seed + (a.rng_seed_dev ? a.rng_seed_dev[0] : 0ull);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint64_t dense = (uint64_t)(d0 + (int64_t)((r & 3) + 8 * (r >> 2)) * T);
                float u0, u1;
                nef_rng_uniform2(seed, dense, u0, u1);          // t and T are even: (dense, dense + 1) is an aligned pair
                y0[r] = (u0 >= a.drop_p) ? y0[r] * a.drop_scale : 0.f;
                y1[r] = (u1 >= a.drop_p) ? y1[r] * a.drop_scale : 0.f;
            }
        }
        if (a.gate) {
            const float* gp = a.gate + (int64_t)b0 * a.gate_bs + (int64_t)g * a.gate_gs + (int64_t)cobase * T + ts;
            f32x2 t16[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                t16[r] = *reinterpret_cast<const f32x2*>(gp + (int64_t)((r & 3) + 8 * (r >> 2)) * T);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                y0[r] = t16[r][0] > 0.f ? y0[r] * a.gate_scale : 0.f;
                y1[r] = t16[r][1] > 0.f ? y1[r] * a.gate_scale : 0.f;
            }
        }
        if (live) {
            float* yp = a.y + (int64_t)b0 * a.y_bs + (int64_t)g * a.y_gs + (int64_t)cobase * T + t;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                f32x2 o;
                o[0] = y0[r];
                o[1] = y1[r];
                *reinterpret_cast<f32x2*>(yp + (int64_t)((r & 3) + 8 * (r >> 2)) * T) = o;
            }
        }
    }
}

template <int K, int WM, int PRO = 0>
static int launch_conv_wino(const nef_conv_args& a, hipStream_t st) {
    constexpr int MT = 64 * WM;
    constexpr int NTO = 64 * (4 / WM);
    constexpr size_t lds = (size_t)(2 * WKC * (NTO + 16) + ((PRO & 1) ? 2 * PRO_MAX_CIN : 0)) * sizeof(float);
    if ((PRO & 1) && a.Cin_g > PRO_MAX_CIN) return NEF_E_SHAPE;
    static unsigned long long lds_set = 0;      // per-device bits, see nef_ensure_dyn_lds
    if (int e = nef_ensure_dyn_lds(reinterpret_cast<const void*>(&conv_wino_kernel<K, WM, PRO>), lds, &lds_set)) return e;
    const int tps = (a.T + NTO - 1) / NTO;
    const int n_tiles = a.B * tps;
    const int m_tiles = a.Cout_g / MT;
    const int64_t blocks = (int64_t)a.G * m_tiles * n_tiles;
    if (blocks <= 0 || blocks > 0x7fffffff) return NEF_E_SHAPE;
    hipLaunchKernelGGL((conv_wino_kernel<K, WM, PRO>), dim3((unsigned)blocks), dim3(256), lds, st, a, tps, n_tiles, m_tiles);
    return nef_launch_status();
}

// ------------------------------------------------------------------------------------------------
// The same family one size up: Winograd F(4,3) -- 6 multiplies per FOUR outputs (1.5 per output; F(2,3): 2, direct: 3)
// ------------------------------------------------------------------------------------------------
// Quad j of a row: outputs 4j..4j+3, inputs d_m = x[4j-1+m], m = 0..5 (points 0, +-1, +-2, inf):
//     v = B^T d:  v0 = 4d0-5d2+d4   v1 = -4d1-4d2+d3+d4   v2 = 4d1-4d2-d3+d4   v3 = -2d1-d2+2d3+d4   v4 = 2d1-d2-2d3+d4
//                 v5 = 4d1-5d3+d5
//     u = G g:    u0 = g0/4   u1 = -(g0+g1+g2)/6   u2 = -(g0-g1+g2)/6   u3 = g0/24+g1/12+g2/6   u4 = g0/24-g1/12+g2/6   u5 = g2
//     M_i[co][j] = sum_ci u_i v_i  (6 GEMMs over a QUARTER of the columns)
//     y0 = M0+M1+M2+M3+M4   y1 = M1-M2+2M3-2M4   y2 = M1+M2+4M3+4M4   y3 = M1-M2+8M3-8M4+M5
// K = 7 (round 3) = taps split 4 + 3: F(4,4) on taps 0..3 and inputs d_m = x[4j-3+m], m = 0..6 (points 0, +-1, +-2, 1/2, inf:
// 7 products), F(4,3) on taps 4..6 and inputs x[4j+1 .. 4j+6].  The output-transform columns of the six points the two
// share are identical, so both groups accumulate into the SAME six M tiles and the point 1/2 gets a seventh:
//     y0 += M6   y1 += M6/2   y2 += M6/4   y3 += M6/8
// 7 + 6 = 13 multiplies per 4 outputs (3.25 per output; rounds 1-2: three F(4,3) groups 3+3+1 = 17, F(2,3) split: 20,
// direct: 28).  B^T of F(4,4) (rows in accumulator order 0, 1, -1, 2, -2, inf, 1/2):
//     (-2, 4, 2.5, -5, -0.5, 1, 0)  (0, 2, -2, -4.5, 0.5, 1, 0)  (0, -2, 6, -3.5, -1.5, 1, 0)  (0, 1, -1.5, -2, 1.5, 1, 0)
//     (0, -1, 2.5, 0, -2.5, 1, 0)   (0, -2, 4, 2.5, -5, -0.5, 1)  (0, 4, 0, -5, 0, 1, 0)
// G rows: -g0/2, -(g0+g1+g2+g3)/3, (g0-g1+g2-g3)/9, g0/36+g1/18+g2/9+2g3/9, -g0/60+g1/30-g2/15+2g3/15, g3, (32g0+16g1+8g2+4g3)/45.
// fp32 rounding (numpy model of the whole pipeline, 128 channels): 1.2x the 3+3+1 F(4,3) form, 1.6x the direct form --
// used for backward-data launches only (no decision is taken on a gradient); the whole-model gradient bars are unchanged.
// Machinery as conv_wino_kernel (raw activations double-buffered in LDS and transformed on the way to the B operand,
// A fragments straight from L2 ahead of use, one barrier per 16-channel stage); a wave owns 32 output channels x 32 quads
// (128 outputs) = 6 accumulator tiles; a workgroup is 4 x 1 waves (128 channels x 128 outputs) or 2 x 2 (64 x 256).
// K = 3, 128 channels, no upsampling prologue: 168 VGPRs -> three workgroups per CU (-3..5 % against two)
#ifndef NEF_W4_MINB3
#define NEF_W4_MINB3 1
#endif
#ifndef NEF_W4_AHEAD3_W2
#define NEF_W4_AHEAD3_W2 3
#endif
#ifndef NEF_W4_AHEAD3_W4
#define NEF_W4_AHEAD3_W4 3
#endif
#ifndef NEF_W4_AHEAD7
#define NEF_W4_AHEAD7 1
#endif
#ifndef NEF_W4_MINB3_W2
#define NEF_W4_MINB3_W2 0
#endif
constexpr int w4_wgs_per_cu(int K, int WMC, int PRO) {
    return (K == 3 && (PRO & 2) == 0 && ((WMC == 4 && NEF_W4_MINB3) || (WMC == 2 && NEF_W4_MINB3_W2))) ? 3 : 2;
}
template <int K, int WMC, int PRO>
__global__ __launch_bounds__(256, w4_wgs_per_cu(K, WMC, PRO)) void conv_wino4_kernel(nef_conv_args a, int tps, int n_tiles, int m_tiles) {
    constexpr bool UP = (PRO & 2) != 0, AFF = (PRO & 1) != 0;
    constexpr int NS = UP ? 2 : 1;
    constexpr int KC = WKC;
    constexpr int WN = 4 / WMC;
    constexpr int PAD = (K - 1) / 2;
    constexpr int NPL = K == 3 ? 6 : 13;     // weight planes per (ci, co)
    constexpr int NACC = K == 3 ? 6 : 7;     // M tiles
    constexpr int MT = 32 * WMC;             // output channels per workgroup
    constexpr int NTO = 128 * WN;            // outputs (columns) per workgroup
    constexpr int NXV = K == 3 ? 3 : 5;      // ds_read_b64 per lane and k-step: x[4j-PAD .. 4j-PAD+2*NXV)
    constexpr int XROW = NTO + 2 * NXV - 4;  // staged positions per channel row
    constexpr int XRS = NTO + 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xl = smem;                        // [2][KC][XRS]
    float* Pl = smem + 2 * KC * XRS;         // [2][Cin_g]: prologue affine of this tile's pass (AFF only)
    // [5][MT]: what the epilogue needs per output channel (bias; mean, invstd, a, b of the BatchNorm below for the bnb sums),
    // fetched while the first tile is in flight -- in the epilogue each of these was a dependent global load with the
    // memory latency of a loaded chip in front of the output stores
    float* El = Pl + (AFF ? 2 * PRO_MAX_CIN : 0);
#ifdef NEF_TRACE
    const bool tr_on = nef_trace_ptr != nullptr && (blockIdx.x & 15) == 0;
    unsigned long long tr_t[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) tr_t[i] = 0;
#endif
    NEF_TR(0)
#ifdef NEF_TRACE
    if (tr_on) tr_t[22] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
#endif

    const int tile = blockIdx.x % n_tiles;
    const int gm = blockIdx.x / n_tiles;
    const int mt = gm % m_tiles;
    const int g = gm / m_tiles;
    int b0, t0;
    {
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
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    constexpr int NIT = (XROW + 63) / 64;
    constexpr int XR = KC / 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm_u = wave_u / WN;
    const float* const xbase = a.x + (int64_t)b0 * a.x_bs + (int64_t)g * a.x_gs;
    const __amdgpu_buffer_rsrc_t xrs = nef_rsrc(xbase);
    constexpr int NQ4 = NPL / 4, REM = NPL % 4;              // 16-byte vectors + an 8- or 4-byte tail per lane and k-step
    const int a_rstride = (Cog >> 5) * 128;                  // floats per reduction channel inside one full-vector slab
    const int a_qstride = Cig * a_rstride;                   // floats per slab; the tail slab follows the NQ4 full ones
    const __amdgpu_buffer_rsrc_t wrs = nef_rsrc(a.wp + (int64_t)g * NPL * Cig * Cog + ((m0 >> 5) + wm_u) * 128);
    const __amdgpu_buffer_rsrc_t wrs_r = nef_rsrc(a.wp + (int64_t)g * NPL * Cig * Cog + (int64_t)NQ4 * a_qstride +
                                                  ((m0 >> 5) + wm_u) * (32 * REM));
    const int Tin = UP ? (T >> 1) : T;
    unsigned xvo[NIT][NS];
    float lam[NIT];
    bool xok[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int r = lane + 64 * it;
        const int t = t0 + r - PAD;
        xok[it] = (r < XROW) && (t >= 0) && (t < T);
        lam[it] = 0.f;
        if constexpr (UP) {
            float src = 0.5f * ((float)t + 0.5f) - 0.5f;
            if (src < 0.f) src = 0.f;
            int i0 = (int)src;
            if (i0 > Tin - 1) i0 = Tin - 1;
            const int i1 = i0 + (i0 < Tin - 1 ? 1 : 0);
            lam[it] = src - (float)i0;
            xvo[it][0] = xok[it] ? (unsigned)(i0 * 4) : NEF_OOB;
            xvo[it][NS - 1] = xok[it] ? (unsigned)(i1 * 4) : NEF_OOB;
        } else {
            xvo[it][0] = xok[it] ? (unsigned)(t * 4) : NEF_OOB;
        }
    }
    const int64_t soff = (int64_t)b0 * a.sc_bs + (int64_t)g * a.sc_gs;
    const int pro_row0 = AFF ? (b0 / a.pro_Bp) * a.G * Cig + g * Cig : 0;
    const int a_rstride_r = (Cog >> 5) * (32 * REM);
    const unsigned avo = (unsigned)((hi * a_rstride + 4 * lo) * 4);
    const unsigned avo_r = (unsigned)((hi * a_rstride_r + REM * lo) * 4);

    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    constexpr int SPK = KC / 2;
    // A ring depth.  Vector-memory results return IN ORDER, so the wait for an A fragment that was requested after the
    // activation rows of the next stage also waits for those rows: the ring depth is the cover (in k-steps) the activation
    // fetch gets before the wave is made to sit on it -- and with it the share of a stage during which a workgroup has
    // bytes in flight at all (the 64-channel layers are bound by exactly that: ~17 KB per workgroup, 40 % of the time).
    constexpr int AHEAD = K == 3 ? (WMC == 2 ? NEF_W4_AHEAD3_W2 : NEF_W4_AHEAD3_W4) : NEF_W4_AHEAD7;
    constexpr int NSET = AHEAD + 1;
    static_assert(SPK % NSET == 0, "the A sets must line up across stages");
    const int nsteps = Cig / 2;
    // A operand: nef_pack_weight_wino4 lays the NPL values a lane needs per k-step out as NQ4 slabs [ci][32-wide co
    // block][lo][4] of full 16-byte vectors (plane 4*q + e) followed by one tail slab [ci][block][lo][REM]: K = 3: one
    // 16-byte + one 8-byte load instead of 6 dwords, K = 7: four 16-byte loads + one dword instead of 17 (slabs outermost:
    // see conv_wino_kernel)
    f32x4 fa4[NSET][NQ4];
    float far[NSET][REM];
#define NEF_FA4(SET, I) ((I) < 4 * NQ4 ? fa4[SET][((I) < 4 * NQ4 ? (I) : 0) >> 2][(I) & 3] : far[SET][(I) >= 4 * NQ4 ? (I) - 4 * NQ4 : 0])
    float xreg[XR][NIT][NS];
#define NEF_W4A_ISSUE(GS, SET)                                                                                       \
    {                                                                                                               \
        const int gs_ = (GS) < nsteps ? (GS) : nsteps - 1;                                                          \
        _Pragma("unroll") for (int q = 0; q < NQ4; ++q)                                                             \
            fa4[SET][q] = nef_buf_f32x4(wrs, avo, (unsigned)((q * a_qstride + 2 * gs_ * a_rstride) * 4));            \
        if constexpr (REM == 2) {                                                                                   \
            const f32x2 t_ = nef_buf_f32x2(wrs_r, avo_r, (unsigned)((2 * gs_ * a_rstride_r) * 4));                  \
            far[SET][0] = t_[0];                                                                                    \
            far[SET][REM - 1] = t_[1];                                                                              \
        } else {                                                                                                    \
            far[SET][0] = nef_buf_f32(wrs_r, avo_r, (unsigned)((2 * gs_ * a_rstride_r) * 4));                       \
        }                                                                                                           \
    }
#define NEF_W4X_ISSUE(C0, RS)                                                                                        \
    {                                                                                                               \
        _Pragma("unroll") for (int rr = 0; rr < XR; ++rr) {                                                         \
            const unsigned so = (unsigned)(((C0) + wave_u + 4 * rr) * Tin * 4);                                     \
            _Pragma("unroll") for (int it = 0; it < NIT; ++it)                                                      \
                _Pragma("unroll") for (int ns = 0; ns < NS; ++ns) xreg[rr][it][ns] = nef_buf_f32(RS, xvo[it][ns], so); \
        }                                                                                                           \
    }
#define NEF_W4X_STORE(C0, BUFP)                                                                                      \
    {                                                                                                               \
        if (a.in_scale) {                                                                                           \
            _Pragma("unroll") for (int rr = 0; rr < XR; ++rr) {                                                     \
                const float sc = a.in_scale[soff + (C0) + wave + 4 * rr];                                           \
                _Pragma("unroll") for (int it = 0; it < NIT; ++it) xreg[rr][it][0] *= sc;                           \
            }                                                                                                       \
        }                                                                                                           \
        _Pragma("unroll") for (int rr = 0; rr < XR; ++rr) {                                                         \
            float pa = 1.f, pb = 0.f;                                                                               \
            if constexpr (AFF) {                                                                                    \
                pa = Pl[(C0) + wave_u + 4 * rr];                                                                    \
                pb = Pl[Cig + (C0) + wave_u + 4 * rr];                                                              \
            }                                                                                                       \
            _Pragma("unroll") for (int it = 0; it < NIT; ++it) {                                                    \
                const int r = lane + 64 * it;                                                                       \
                float v = xreg[rr][it][0];                                                                          \
                if constexpr (AFF) v = fmaxf(fmaf(v, pa, pb), 0.f);                                                 \
                if constexpr (UP) {                                                                                 \
                    float v1 = xreg[rr][it][NS - 1];                                                                \
                    if constexpr (AFF) v1 = fmaxf(fmaf(v1, pa, pb), 0.f);                                           \
                    v = (1.f - lam[it]) * v + lam[it] * v1;                                                         \
                }                                                                                                   \
                if constexpr (PRO != 0) v = xok[it] ? v : 0.f;                                                      \
                if (r < XROW) (BUFP)[(wave + 4 * rr) * XRS + r] = v;                                                \
            }                                                                                                       \
        }                                                                                                           \
    }
    NEF_W4X_ISSUE(0, xrs)
#pragma unroll
    for (int s_ = 0; s_ < ((NEF_ABL & 1) ? NSET : AHEAD); ++s_) NEF_W4A_ISSUE(s_, s_)
    if (threadIdx.x < MT) {     // published by the barrier behind the first tile's LDS stores
        const int ch_ = g * Cog + m0 + (int)threadIdx.x;
        El[threadIdx.x] = a.bias ? a.bias[ch_] : 0.f;
        if (a.bnb_slots) {
            const int pr_ = (b0 / a.bnb_Bp) * a.G * Cog + ch_;
            El[MT + threadIdx.x] = a.bnb_mean[pr_];
            El[2 * MT + threadIdx.x] = a.bnb_invstd[pr_];
            El[3 * MT + threadIdx.x] = a.bnb_a[pr_];
            El[4 * MT + threadIdx.x] = a.bnb_b[pr_];
        }
    }
    if constexpr (AFF) {     // see conv_wino_kernel
        for (int i = threadIdx.x; i < Cig; i += 256) {
            Pl[i] = a.pro_a[pro_row0 + i];
            Pl[Cig + i] = a.pro_b[pro_row0 + i];
        }
        __syncthreads();
    }
    NEF_TR(1)
    NEF_W4X_STORE(0, Xl)
    __syncthreads();
    NEF_TR(2)
    int st = 0;
    f32x2 fx[2][NXV];
#define NEF_W4X_LOAD(S, BUF)                                                                                         \
    {                                                                                                               \
        const f32x2* xp_ = reinterpret_cast<const f32x2*>(xb + 2 * (S) * XRS);                                      \
        _Pragma("unroll") for (int q_ = 0; q_ < NXV; ++q_) fx[BUF][q_] = xp_[q_];                                   \
    }
    if constexpr ((NEF_ABL & 4) != 0) {
        const float* xb = Xl + hi * XRS + 4 * (wn * 32 + lo);
        NEF_W4X_LOAD(0, 0)
        NEF_W4X_LOAD(1, 1)
    }
    for (int c0 = 0; c0 < Cig; c0 += KC, ++st) {
        const float* xb = Xl + ((NEF_ABL & 2) ? 0 : (st & 1)) * (KC * XRS) + hi * XRS + 4 * (wn * 32 + lo);
        const bool more = c0 + KC < Cig;
        const __amdgpu_buffer_rsrc_t xrs_n = nef_rsrc_n(xbase, more ? 0x7FFFFFFCu : 0u);     // see conv_wino_kernel
        if constexpr (!(NEF_ABL & 4)) NEF_W4X_LOAD(0, 0)
#pragma unroll
        for (int s_ = 0; s_ < SPK; ++s_) {
            if constexpr (!(NEF_ABL & 1)) NEF_W4A_ISSUE(st * SPK + s_ + AHEAD, (s_ + AHEAD) % NSET)
            if constexpr (!(NEF_ABL & 2)) if (s_ == 0) NEF_W4X_ISSUE(c0 + KC, xrs_n)
            if constexpr (!(NEF_ABL & 4)) if (s_ + 1 < SPK) NEF_W4X_LOAD(s_ + 1, (s_ + 1) & 1)
            __builtin_amdgcn_s_setprio(1);      // scheduling fence, see conv_wino_kernel
            float x_[2 * NXV];
#pragma unroll
            for (int q_ = 0; q_ < NXV; ++q_) {
                x_[2 * q_] = fx[s_ & 1][q_][0];
                x_[2 * q_ + 1] = fx[s_ & 1][q_][1];
            }
            if constexpr (K == 7) {     // F(4,4) on taps 0..3, inputs x_[0..6]; planes 0..6 in accumulator order
                const float d0 = x_[0], d1 = x_[1], d2 = x_[2], d3 = x_[3], d4 = x_[4], d5 = x_[5], d6 = x_[6];
                float v[7];
                v[6] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));                                   // point 1/2
                v[0] = fmaf(-2.f, d0, fmaf(2.5f, d2, fmaf(-0.5f, d4, v[6])));
                v[5] = fmaf(-2.f, d1, fmaf(2.5f, d3, fmaf(-0.5f, d5, fmaf(4.f, d2, fmaf(-5.f, d4, d6)))));   // infinity
                const float p = fmaf(2.f, d2, fmaf(-4.f, d3, fmaf(-0.5f, d4, d5)));
                const float q = fmaf(2.f, d1, fmaf(-4.f, d2, fmaf(-0.5f, d3, d4)));
                v[1] = p + q;
                v[2] = p - q;
                const float d24 = d2 - d4;
                const float p2 = fmaf(0.5f, d24, d5 - d3), q2 = fmaf(-2.f, d24, d1 - d3);
                v[3] = p2 + q2;
                v[4] = p2 - q2;
#pragma unroll
                for (int i = 0; i < 7; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(NEF_FA4(s_ % NSET, i), v[i], acc[i], 0, 0, 0);
            }
            {                           // F(4,3): K = 3 on x_[0..5]; K = 7 on taps 4..6, inputs x_[4..9], planes 7..12
                constexpr int X0 = K == 3 ? 0 : 4, P0 = K == 3 ? 0 : 7;
                const float d0 = x_[X0], d1 = x_[X0 + 1], d2 = x_[X0 + 2], d3 = x_[X0 + 3], d4 = x_[X0 + 4], d5 = x_[X0 + 5];
                float v[6];
                const float t1 = fmaf(-4.f, d2, d4), t2 = fmaf(-4.f, d1, d3);
                const float t3 = d4 - d2, t4 = 2.f * (d3 - d1);
                v[0] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
                v[1] = t1 + t2;
                v[2] = t1 - t2;
                v[3] = t3 + t4;
                v[4] = t3 - t4;
                v[5] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(NEF_FA4(s_ % NSET, P0 + i), v[i], acc[i], 0, 0, 0);
            }
        }
#ifdef NEF_TRACE
        if (tr_on) {      // stage st: [3 + 2*st] = MFMA loop done, [4 + 2*st] = next tile stored + barrier passed
#pragma unroll
            for (int q_ = 0; q_ < 8; ++q_) if (q_ == st) tr_t[3 + 2 * q_] = __builtin_readcyclecounter();
        }
#endif
        if constexpr (!(NEF_ABL & 2)) if (more) NEF_W4X_STORE(c0 + KC, Xl + ((st + 1) & 1) * (KC * XRS))
        __syncthreads();
#ifdef NEF_TRACE
        if (tr_on) {
#pragma unroll
            for (int q_ = 0; q_ < 8; ++q_) if (q_ == st) tr_t[4 + 2 * q_] = __builtin_readcyclecounter();
        }
#endif
    }
#undef NEF_W4X_LOAD
#undef NEF_W4A_ISSUE
#undef NEF_FA4
#undef NEF_W4X_ISSUE
#undef NEF_W4X_STORE
#ifdef NEF_TRACE
    if constexpr ((NEF_ABL & 8) != 0) {
        if (tr_on && threadIdx.x == 0) {
            unsigned long long* o = nef_trace_ptr + (size_t)(blockIdx.x >> 4) * 24;
            tr_t[20] = tr_t[21] = __builtin_readcyclecounter();
#pragma unroll
            for (int i = 0; i < 24; ++i) o[i] = tr_t[i];
        }
    }
#endif
    if constexpr ((NEF_ABL & 8) != 0) if (a.T >= 0) return;      // run-time true: the epilogue below is dead at run time only

    // epilogue: output transform, then the usual bias / residual / ReLU / dropout / gate on the four adjacent outputs a
    // lane owns per channel row (two 8-byte accesses; T is even, so each pair is inside or outside the row as a whole)
    const int64_t ctot = (int64_t)a.G * Cog;
    const int t = t0 + 4 * (wn * 32 + lo);
    const bool inb = b0 < a.B;
    const bool live[2] = {inb && t < T, inb && t + 2 < T};
    const int ts[2] = {live[0] ? t : 0, live[1] ? t + 2 : 0};
    const int cobase = m0 + wm * 32 + 4 * hi;
#define NEF_ROW4(q, h) ((((q) + 8 * (h)) & 3) + 8 * (((q) + 8 * (h)) >> 2))
    float sv[32];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#define NEF_ROW(q) ((((q) + 8 * h) & 3) + 8 * (((q) + 8 * h) >> 2))
        float y[8][4];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = q + 8 * h;
            const float m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r], m4 = acc[4][r];
            const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
            y[q][0] = (acc[0][r] + s12) + s34;
            y[q][1] = fmaf(2.f, d34, d12);
            y[q][2] = fmaf(4.f, s34, s12);
            y[q][3] = fmaf(8.f, d34, d12) + acc[5][r];
            if constexpr (K == 7) {      // the F(4,4) group's point 1/2
                const float m6 = acc[NACC - 1][r];
                y[q][0] += m6;
                y[q][1] = fmaf(0.5f, m6, y[q][1]);
                y[q][2] = fmaf(0.25f, m6, y[q][2]);
                y[q][3] = fmaf(0.125f, m6, y[q][3]);
            }
        }
        if (a.bias) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float bv = El[wm * 32 + 4 * hi + NEF_ROW(q)];
#pragma unroll
                for (int e = 0; e < 4; ++e) y[q][e] += bv;
            }
        }
        // residual and gate operands: 16-byte fetches (one per row instead of two 8-byte ones) on every tile that lies
        // inside the row; the ragged last tile of a row keeps the pair loads (safe coordinates for dead lanes).  Each
        // operand is consumed right after its fetch, so only one of them occupies registers at a time.
        const bool ragged = t0 + NTO > T;                    // workgroup-uniform
#define NEF_EPI_FETCH4(PTR, BS, GS, DST)                                                                              \
    if (!ragged) {                                                                                                  \
        const __amdgpu_buffer_rsrc_t rs_ = nef_rsrc((PTR) + (int64_t)b0 * (BS) + (int64_t)g * (GS) + (int64_t)(m0 + wm * 32) * T); \
        const unsigned vo_ = inb ? (unsigned)((4 * hi * T + t) * 4) : NEF_OOB;                                      \
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
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) y[q][e] += rv[q][e];
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
                const uint8_t* mp = a.mask + ((int64_t)b0 * ctot + (int64_t)g * Cog + cobase) * T + ts[pr];
                unsigned short t8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) t8[q] = *reinterpret_cast<const unsigned short*>(mp + (int64_t)NEF_ROW(q) * T);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    y[q][2 * pr] *= (float)(t8[q] & 0xff) * a.drop_scale;
                    y[q][2 * pr + 1] *= (float)(t8[q] >> 8) * a.drop_scale;
                }
            } else if (a.drop_p > 0.f) {
                const int64_t d0 = ((int64_t)b0 * ctot + (int64_t)g * Cog + cobase) * T + ts[pr];
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
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) y[q][e] = gv[q][e] > 0.f ? y[q][e] * a.gate_scale : 0.f;
        }
#undef NEF_EPI_FETCH4
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            // a quad that ends inside the row is stored below as ONE 16-byte vector; only the half-live quad at the end of
            // a row with T % 4 == 2 goes out as a pair
            if (live[pr] && !live[1]) {
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
        {   // output stores: the epilogue is the largest single cost of the K = 3 launches (timing-only builds: 14..29 %), and it
            // is bound by store ISSUE, not bandwidth -- one buffer_store_dwordx4 per row (512 contiguous bytes per half-wave)
            // instead of two 8-byte stores; lanes outside the row carry NEF_OOB and store nothing
            const __amdgpu_buffer_rsrc_t yrs =
                nef_rsrc(a.y + (int64_t)b0 * a.y_bs + (int64_t)g * a.y_gs + (int64_t)(m0 + wm * 32) * T);
            const unsigned yvo = live[1] ? (unsigned)((4 * hi * T + t) * 4) : NEF_OOB;
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
#undef NEF_ROW
        if (a.bnb_slots && a.bnb_up) {
            // ... with a x2 upsampling between that layer and this launch's output: sum_t' m[t'] (U^T g)[t'] = sum_t g[t] (U m)[t],
            // so the lane weighs its four outputs t = 4j..4j+3 with the upsampled decision (and decision * xhat) rows,
            // built from the half-resolution tile x[2j-1 .. 2j+2] (indices clamped as nn.Upsample clamps them)
            const int Lh = T >> 1;
            const float* xp = a.bnb_x + ((int64_t)b0 * ctot + (int64_t)g * Cog + cobase) * Lh;
            const int j2 = live[0] ? (t >> 1) : 0;
            const int im1 = j2 > 0 ? j2 - 1 : 0, i1 = j2 + 1 < Lh ? j2 + 1 : Lh - 1, ip2 = j2 + 2 < Lh ? j2 + 2 : Lh - 1;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int row = NEF_ROW4(q, h);
                const int er = wm * 32 + 4 * hi + row;
                const float af = El[3 * MT + er], bf = El[4 * MT + er];
                const float mf = El[MT + er], is = El[2 * MT + er];
                const float* xr = xp + (int64_t)row * Lh;
                const float xa = xr[im1], xb = xr[j2], xc = xr[i1], xd = xr[ip2];
                const float ma = fmaf(xa, af, bf) > 0.f ? 1.f : 0.f, mb = fmaf(xb, af, bf) > 0.f ? 1.f : 0.f;
                const float mc = fmaf(xc, af, bf) > 0.f ? 1.f : 0.f, md = fmaf(xd, af, bf) > 0.f ? 1.f : 0.f;
                const float ha = ma * ((xa - mf) * is), hb = mb * ((xb - mf) * is);
                const float hc = mc * ((xc - mf) * is), hd = md * ((xd - mf) * is);
                const float g0 = live[0] ? y[q][0] : 0.f, g1 = live[0] ? y[q][1] : 0.f;
                const float g2 = live[1] ? y[q][2] : 0.f, g3 = live[1] ? y[q][3] : 0.f;
                sv[2 * (q + 8 * h)] = fmaf(g0, fmaf(0.75f, mb, 0.25f * ma), g1 * fmaf(0.75f, mb, 0.25f * mc)) +
                                      fmaf(g2, fmaf(0.75f, mc, 0.25f * mb), g3 * fmaf(0.75f, mc, 0.25f * md));
                sv[2 * (q + 8 * h) + 1] = fmaf(g0, fmaf(0.75f, hb, 0.25f * ha), g1 * fmaf(0.75f, hb, 0.25f * hc)) +
                                          fmaf(g2, fmaf(0.75f, hc, 0.25f * hb), g3 * fmaf(0.75f, hc, 0.25f * hd));
            }
        } else if (a.bnb_slots) {   // BatchNorm-backward sums of the layer below: g*m and g*m*xhat over this lane's live outputs
            const float* xp = a.bnb_x + ((int64_t)b0 * ctot + (int64_t)g * Cog + cobase) * T;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int row = NEF_ROW4(q, h);
                const int er = wm * 32 + 4 * hi + row;
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
        } else if (a.stats) {   // this lane's share of the slot sums: its (up to) four live outputs of each of its 8 rows
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float y0 = live[0] ? y[q][0] : 0.f, y1 = live[0] ? y[q][1] : 0.f;
                const float y2 = live[1] ? y[q][2] : 0.f, y3 = live[1] ? y[q][3] : 0.f;
                sv[2 * (q + 8 * h)] = (y0 + y1) + (y2 + y3);
                sv[2 * (q + 8 * h) + 1] = fmaf(y0, y0, y1 * y1) + fmaf(y2, y2, y3 * y3);
            }
        }
    }
    float* const slot_out = a.bnb_slots ? a.bnb_slots : a.stats;
    if (slot_out) {
        // 32 values (16 rows x {sum, sum of squares}) summed over the 32 lanes that share `hi`: a halving butterfly -- each
        // step a lane passes on the half it does not keep, 16+8+4+2+1 shuffles instead of 32 x 5 -- after which lane `lo`
        // holds the slot total of value `lo` = 2*r + {0,1}, row r = q + 8h.  Fixed order: deterministic.
#pragma unroll
        for (int step = 0; step < 5; ++step) {
            const int off = 16 >> step;
            const bool up = (lo & off) != 0;
#pragma unroll
            for (int i = 0; i < off; ++i) {
                const float send = up ? sv[i] : sv[i + off];
                const float keep = up ? sv[i + off] : sv[i];
                sv[i] = keep + __shfl_xor(send, off, 64);
            }
        }
        const int r = lo >> 1;
        const int ch = g * Cog + cobase + (r & 3) + 8 * (r >> 2);
        const int64_t nslot = (int64_t)tps * WN;
        const int64_t slot = (int64_t)b0 * nslot + (int64_t)(t0 / NTO) * WN + wn;
#undef NEF_ROW4
        if (inb) slot_out[((int64_t)ch * a.B * nslot + slot) * 2 + (lo & 1)] = sv[0];
    }
#ifdef NEF_TRACE
    NEF_TR(20)
    if (tr_on) {
        __builtin_amdgcn_s_waitcnt(0);      // vmcnt(0): the output stores have left the wave's queue
        tr_t[21] = __builtin_readcyclecounter();
        if (threadIdx.x == 0) {
            unsigned long long* o = nef_trace_ptr + (size_t)(blockIdx.x >> 4) * 24;
#pragma unroll
            for (int i = 0; i < 24; ++i) o[i] = tr_t[i];
        }
    }
#endif
}

template <int K, int WMC, int PRO = 0>
static int launch_conv_wino4(const nef_conv_args& a, hipStream_t st) {
    constexpr int MT = 32 * WMC;
    constexpr int NTO = 128 * (4 / WMC);
    constexpr size_t lds = (size_t)(2 * WKC * (NTO + 16) + ((PRO & 1) ? 2 * PRO_MAX_CIN : 0) + 5 * MT) * sizeof(float);
    if ((PRO & 1) && a.Cin_g > PRO_MAX_CIN) return NEF_E_SHAPE;
    static unsigned long long lds_set = 0;      // per-device bits, see nef_ensure_dyn_lds
#ifdef NEF_TRACE
    size_t lds_launch = lds;       // NEF_DEBUG_LDS=<bytes>: inflate the LDS request to force fewer workgroups per CU
    if (const char* e_ = nef_diag_env("NEF_DEBUG_LDS")) lds_launch = (size_t)atol(e_);
    if (int e = nef_ensure_dyn_lds(reinterpret_cast<const void*>(&conv_wino4_kernel<K, WMC, PRO>), 160 * 1024, &lds_set)) return e;
#else
    constexpr size_t lds_launch = lds;
    if (int e = nef_ensure_dyn_lds(reinterpret_cast<const void*>(&conv_wino4_kernel<K, WMC, PRO>), lds, &lds_set)) return e;
#endif
    const int tps = (a.T + NTO - 1) / NTO;
    const int n_tiles = a.B * tps;
    const int m_tiles = a.Cout_g / MT;
    const int64_t blocks = (int64_t)a.G * m_tiles * n_tiles;
    if (blocks <= 0 || blocks > 0x7fffffff) return NEF_E_SHAPE;
    hipLaunchKernelGGL((conv_wino4_kernel<K, WMC, PRO>), dim3((unsigned)blocks), dim3(256), lds_launch, st, a, tps, n_tiles, m_tiles);
    return nef_launch_status();
}

// Operand of conv_wino4_kernel: per group NPL / 4 slabs [r][32-wide block of c][lo][4] (plane pl = 4*q + e) followed by
// one tail slab [r][block][lo][NPL % 4]; (r, c) = (ci, co) forward, (co, ci) with the taps reversed for the backward-data
// operand, lo = c % 32.  Planes: K = 3: 0..5 = G (g0,g1,g2) of F(4,3).  K = 7: 0..6 = G of F(4,4) applied to taps 0..3
// (accumulator order: points 0, 1, -1, 2, -2, inf, 1/2), 7..12 = G of F(4,3) applied to taps 4..6.
__device__ __forceinline__ void pack_wino4_elem(const float* __restrict__ w, float* __restrict__ wp, int G, int Cog,
                                                int Cig, int K, int flip, int64_t i) {
    const int npl = K == 3 ? 6 : 13;
    int64_t q = i;
    int co, ci;
    if (!flip) {
        co = (int)(q % Cog); q /= Cog;
        ci = (int)(q % Cig); q /= Cig;
    } else {
        ci = (int)(q % Cig); q /= Cig;
        co = (int)(q % Cog); q /= Cog;
    }
    const int g = (int)q;
    const int r = flip ? co : ci, c = flip ? ci : co, Cr = flip ? Cog : Cig, Cc = flip ? Cig : Cog;
    const float* src = w + (((int64_t)g * Cog + co) * Cig + ci) * K;
    const int nq4 = npl / 4, rem = npl % 4, lo = c & 31;
    float* const gbase = wp + (int64_t)g * npl * Cr * Cc;
    const int64_t blk = (int64_t)r * (Cc >> 5) + (c >> 5), qstride = (int64_t)Cr * (Cc >> 5) * 128;
#define NEF_PUT4(PL, VAL)                                                                                             \
    {                                                                                                                \
        const int pl_ = (PL);                                                                                        \
        if (pl_ < 4 * nq4) gbase[(pl_ >> 2) * qstride + blk * 128 + lo * 4 + (pl_ & 3)] = (VAL);                     \
        else gbase[nq4 * qstride + blk * (32 * rem) + lo * rem + (pl_ - 4 * nq4)] = (VAL);                           \
    }
#define NEF_TAP(k) src[flip ? K - 1 - (k) : (k)]
    int p0 = 0, t0 = 0;      // first plane / first tap of the F(4,3) group
    if (K == 7) {
        const float g0 = NEF_TAP(0), g1 = NEF_TAP(1), g2 = NEF_TAP(2), g3 = NEF_TAP(3);
        NEF_PUT4(0, g0 * -0.5f)
        NEF_PUT4(1, ((g0 + g1) + (g2 + g3)) * (-1.0f / 3.0f))
        NEF_PUT4(2, ((g0 - g1) + (g2 - g3)) * (1.0f / 9.0f))
        NEF_PUT4(3, (g0 * (1.0f / 36.0f) + g1 * (1.0f / 18.0f)) + (g2 * (1.0f / 9.0f) + g3 * (2.0f / 9.0f)))
        NEF_PUT4(4, (g1 * (1.0f / 30.0f) - g0 * (1.0f / 60.0f)) + (g3 * (2.0f / 15.0f) - g2 * (1.0f / 15.0f)))
        NEF_PUT4(5, g3)
        NEF_PUT4(6, (g0 * (32.0f / 45.0f) + g1 * (16.0f / 45.0f)) + (g2 * (8.0f / 45.0f) + g3 * (4.0f / 45.0f)))
        p0 = 7;
        t0 = 4;
    }
    {
        const float g0 = NEF_TAP(t0), g1 = NEF_TAP(t0 + 1), g2 = NEF_TAP(t0 + 2);
        const float s02 = g0 + g2;
        NEF_PUT4(p0, g0 * 0.25f)
        NEF_PUT4(p0 + 1, (s02 + g1) * (-1.0f / 6.0f))
        NEF_PUT4(p0 + 2, (s02 - g1) * (-1.0f / 6.0f))
        NEF_PUT4(p0 + 3, (g0 * (1.0f / 24.0f) + g2 * (1.0f / 6.0f)) + g1 * (1.0f / 12.0f))
        NEF_PUT4(p0 + 4, (g0 * (1.0f / 24.0f) + g2 * (1.0f / 6.0f)) - g1 * (1.0f / 12.0f))
        NEF_PUT4(p0 + 5, g2)
    }
#undef NEF_TAP
#undef NEF_PUT4
}

__global__ void pack_weight_wino4_kernel(const float* __restrict__ w, float* __restrict__ wp, int G, int Cog, int Cig,
                                         int K, int flip) {
    const int64_t n = (int64_t)G * Cog * Cig;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        pack_wino4_elem(w, wp, G, Cog, Cig, K, flip, i);
}

// Operand of conv_wino_kernel.  K = 3: [g][plane][r][c].  K = 7: [g][q][r][64-wide block of c][lo][4], inside a block
// c = 32*tm + lo and value 2*plane + tm = 4*q + e.  (r, c) = (ci, co) forward, (co, ci) with the taps reversed for the
// backward-data operand.
// K = 3: planes 0..3 = the F(2,3) weight transform (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2) of the three taps.
// K = 7: planes 0..3 / 4..7 = that transform of taps 0..2 / 3..5, plane 8 = tap 6, plane 9 = -tap 6.
__device__ __forceinline__ void pack_wino_elem(const float* __restrict__ w, float* __restrict__ wp, int G, int Cog,
                                               int Cig, int K, int flip, int64_t i) {
    const int npl = K == 3 ? 4 : 10;
    int64_t q = i;
    int co, ci;
    if (!flip) {   // i enumerates [g][ci][co]
        co = (int)(q % Cog); q /= Cog;
        ci = (int)(q % Cig); q /= Cig;
    } else {       // [g][co][ci]
        ci = (int)(q % Cig); q /= Cig;
        co = (int)(q % Cog); q /= Cog;
    }
    const int g = (int)q;
    const int r = flip ? co : ci, c = flip ? ci : co, Cr = flip ? Cog : Cig, Cc = flip ? Cig : Cog;
    const float* src = w + (((int64_t)g * Cog + co) * Cig + ci) * K;
    const int tm = (c >> 5) & 1, lo = c & 31;
    const int64_t qstride = K == 7 ? (int64_t)Cr * (Cc >> 6) * 128 : (int64_t)Cr * Cc;
    float* blk = wp + (int64_t)g * npl * Cr * Cc + (K == 7 ? ((int64_t)r * (Cc >> 6) + (c >> 6)) * 128 + lo * 4 : (int64_t)r * Cc + c);
#define NEF_PUT(PL, VAL) blk[K == 7 ? ((2 * (PL) + tm) >> 2) * qstride + ((2 * (PL) + tm) & 3) : (PL) * qstride] = (VAL);
#define NEF_TAP(k) src[flip ? K - 1 - (k) : (k)]
    if (K == 3) {
        const float g0 = NEF_TAP(0), g1 = NEF_TAP(1), g2 = NEF_TAP(2);
        NEF_PUT(0, g0)
        NEF_PUT(1, ((g0 + g1) + g2) * 0.5f)
        NEF_PUT(2, ((g0 - g1) + g2) * 0.5f)
        NEF_PUT(3, g2)
    } else {
        // taps 0..3: G of F(2,4) (points 0, 1, -1, 2, inf); taps 4..6: G of F(2,3) with its last plane negated (it accumulates
        // into the F(2,4) group's infinity tile, whose output column is (0, +1) where F(2,3)'s is (0, -1))
        const float g0 = NEF_TAP(0), g1 = NEF_TAP(1), g2 = NEF_TAP(2), g3 = NEF_TAP(3);
        NEF_PUT(0, g0 * 0.5f)
        NEF_PUT(1, ((g0 + g1) + (g2 + g3)) * -0.5f)
        NEF_PUT(2, ((g1 - g0) + (g3 - g2)) * (1.0f / 6.0f))
        NEF_PUT(3, (g0 * (1.0f / 6.0f) + g1 * (1.0f / 3.0f)) + (g2 * (2.0f / 3.0f) + g3 * (4.0f / 3.0f)))
        NEF_PUT(4, g3)
        const float h0 = NEF_TAP(4), h1 = NEF_TAP(5), h2 = NEF_TAP(6);
        NEF_PUT(5, h0)
        NEF_PUT(6, ((h0 + h1) + h2) * 0.5f)
        NEF_PUT(7, ((h0 - h1) + h2) * 0.5f)
        NEF_PUT(8, -h2)
        NEF_PUT(9, 0.f)          // the tenth plane of the 16-byte layout is padding
    }
#undef NEF_TAP
#undef NEF_PUT
}

__global__ void pack_weight_wino_kernel(const float* __restrict__ w, float* __restrict__ wp, int G, int Cog, int Cig,
                                        int K, int flip) {
    const int64_t n = (int64_t)G * Cog * Cig;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        pack_wino_elem(w, wp, G, Cog, Cig, K, flip, i);
}

// ------------------------------------------------------------------------------------------------
// bwd-weight: split over the (b,t) reduction, partials reduced by a second deterministic kernel
// ------------------------------------------------------------------------------------------------
// WINO = 1 (K = 3 or 7, whole 64-column tiles of one sample, T even): the same reduction through the transposed
// Winograd algorithm F(3,2).  For output pair j of a row, g = (gy[2j], gy[2j+1]) and d_m = x[2j-PAD+m]:
//     gW[k] += g0*d_k + g1*d_(k+1), k = 0..2      is      gW = G^T [ (A g) (.) (B^T d) ]
//     A g = (g0, g0+g1, g0-g1, -g1)     B^T d = (d0-d2, d1+d2, d2-d1, d1-d3)     (the forward pass's input transform)
//     M_i[co][ci] = sum over pairs and samples of (A g)_i (B^T d)_i               (4 GEMMs instead of 3, over HALF the columns)
//     gW[0] = M0 + (M1+M2)/2     gW[1] = (M1-M2)/2     gW[2] = (M1+M2)/2 + M3     (once per workgroup, before the partials
// are written; the kernel keeps u3 = +g1 and flips the sign there).  K = 7 splits the taps 3 + 3 + 1 like the forward
// kernel: two transform groups (8 accumulator tiles) plus one plain tile for tap 6: 10 MFMAs per 4 columns instead of 14.
// Both operands are transformed on the way from LDS to the MFMA operands (aligned ds_read_b64 of a lane's own row).
template <int K, int WCO, int TCI, int PRO, int WINO = 0>
__global__ __launch_bounds__(256, 2) void conv_bwd_weight_kernel(
    const float* __restrict__ x, int64_t x_bs, int64_t x_gs, const float* __restrict__ in_scale, int64_t sc_bs,
    int64_t sc_gs, const float* __restrict__ gy, int64_t gy_bs, int64_t gy_gs, float* __restrict__ ws, int B, int T,
    int G, int Cig, int Cog, int seg_shift, int nseg, int tps, int n_tiles, int m_tiles, int ci_chunks, int S,
    const float* __restrict__ pro_a, const float* __restrict__ pro_b, int pro_Bp) {
    constexpr bool UP = (PRO & 2) != 0, AFF = (PRO & 1) != 0;   // same input prologue as conv_fwd_kernel
    constexpr int NS = UP ? 2 : 1;
    const int Tin = UP ? (T >> 1) : T;
    constexpr int WCI = 4 / WCO;
    constexpr int MT = 32 * WCO;
    // WINO = 4 / 5 (K = 7, round 3): the seven taps are split 4 + 3 ACROSS TWO LAUNCHES of this kernel -- WINO = 4 computes taps
    // 0..3 through the transposed F(4,4) (7 products per column quad, 7 accumulator tiles per wave), WINO = 5 taps 4..6
    // through the transposed F(3,4) (6 products, 6 tiles): 13 matrix multiplies per 8 columns where the 4 + 3 split inside
    // one wave (WINO = 3: F(4,2) + F(3,2), 9 tiles) issues 18 and the direct form 28.  One wave cannot hold the 13 tiles.
    // Built and measured slower: the split by WAVE PAIR inside a 64 co x 32 ci workgroup (1.77 vs 1.37 ms: half the matrix
    // work between two barriers for the same staging) and by WORKGROUP TWINS inside one launch (two bodies behind a
    // workgroup-uniform branch: 256 VGPRs + 280 bytes of scratch, whose reloads serialise behind the prefetched tile).
    // gW = G^T [ (A^T gy) (.) (B^T d) ] with the F(4,4) matrices of conv_wino4_kernel.
    constexpr int CIT = 32 * TCI * WCI;
    constexpr int PAD = (K - 1) / 2;
    // LDS row pitches: odd for the transposed ds_read_b32 of the direct form; = 2 (mod 32) for the ds_read_b64 of WINO
    constexpr int GYS = WINO ? 66 : WT + 1;
    constexpr int XS = WINO ? (K == 3 ? 66 : 98) : ((WT + (WT / 16) * (K - 1)) | 1);
    constexpr int NACC = WINO == 4 ? 7 : (WINO == 2 || WINO == 5) ? 6 : (WINO ? (K == 3 ? 4 : 9) : K);
    static_assert(WINO != 2 || K == 3, "the F(3,4) form is for three taps");
    static_assert(WINO != 3 || K == 7, "WINO = 3 is the 4 + 3 split of seven taps");
    static_assert((WINO != 4 && WINO != 5) || (K == 7 && TCI == 1 && PRO == 0), "WINO = 4 / 5 are the two halves of the K = 7 tap split");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* GYl = smem;               // [MT][GYS]
    float* Xl = smem + MT * GYS;     // [CIT][XS]

    // Workgroup -> (unit, ci chunk).  The ci chunks of one unit read the SAME gY tiles; workgroup ids go round-robin over
    // the 8 XCDs (each with its own L2), so the chunks of a unit are placed 8 ids apart: same XCD, dispatched together,
    // and all but the first find the gY tiles in that L2.
    int bid = blockIdx.x;
    int cc;
    {
        const int units = (int)gridDim.x / ci_chunks;
        const int full = (units / 8) * 8 * ci_chunks;          // ids covered by whole groups of 8 units
        if (bid < full) {
            const int grp = bid / (8 * ci_chunks), r = bid % (8 * ci_chunks);
            cc = r / 8;
            bid = grp * 8 + (r % 8);
        } else {
            const int r = bid - full;
            cc = r % ci_chunks;
            bid = (units / 8) * 8 + r / ci_chunks;
        }
    }
    const int mt = bid % m_tiles;
    bid /= m_tiles;
    const int g = bid % G;
    const int split = bid / G;
    const int m0 = mt * MT, c0 = cc * CIT;
    const int seg = 1 << seg_shift;
    const int segw = seg + K - 1;
    const int xrow = nseg * segw;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    const int wco = wave % WCO, wci = wave / WCO;

    f32x16 acc[TCI][NACC];
#pragma unroll
    for (int i = 0; i < TCI; ++i)
#pragma unroll
        for (int k = 0; k < NACC; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;

    // Register-staged pipeline over this workgroup's tiles: loads of tile i+1 fly during the MFMAs of tile i.
    constexpr int GR = MT / 4;        // gY rows per wave
    constexpr int XRW = CIT / 4;      // X rows per wave
    float greg[GR];
    float xreg[XRW][NS];
    // halo positions (beyond 64) per thread, upper bound: the Winograd forms only run on whole tiles of one sample (one
    // segment, K - 1 halo positions per row); the direct form also packs up to four short samples into a tile
    constexpr int NH = WINO ? (CIT * (K - 1) + 255) / 256 : (CIT * ((WT / 16) * (K - 1)) + 255) / 256;
    float xh[NH > 0 ? NH : 1][NS];
    // per-position state of the tile held in registers (needed again when the prologue is applied at the LDS store)
    bool xk_m = false, xk_h[NH > 0 ? NH : 1];
    float lam_m = 0.f, lam_h[NH > 0 ? NH : 1];
    int pass_ld = 0;
    const int nh = xrow - WT;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // source taps of output-resolution time t: (i0, i1, weight of i1); identity when not upsampling
#define NEF_BW_TAPS(TT, I0, I1, LAM)                                                                                 \
    int I0 = (TT), I1 = (TT);                                                                                       \
    float LAM = 0.f;                                                                                                \
    if constexpr (UP) {                                                                                             \
        float src_ = 0.5f * ((float)(TT) + 0.5f) - 0.5f;                                                            \
        if (src_ < 0.f) src_ = 0.f;                                                                                 \
        I0 = (int)src_;                                                                                             \
        if (I0 > Tin - 1) I0 = Tin - 1;                                                                             \
        I1 = I0 + (I0 < Tin - 1 ? 1 : 0);                                                                           \
        LAM = src_ - (float)I0;                                                                                     \
    }
#define NEF_BW_ISSUE(TILE)                                                                                            \
    {                                                                                                               \
        int b0, t0;                                                                                                 \
        if (nseg == 1) { b0 = (TILE) / tps; t0 = ((TILE) - b0 * tps) * WT; } else { b0 = (TILE) * nseg; t0 = 0; }     \
        pass_ld = AFF ? b0 / pro_Bp : 0;                                                                            \
        const __amdgpu_buffer_rsrc_t grs = nef_rsrc(gy + (int64_t)b0 * gy_bs + (int64_t)g * gy_gs + (int64_t)m0 * T); \
        const __amdgpu_buffer_rsrc_t xrs = nef_rsrc(x + (int64_t)b0 * x_bs + (int64_t)g * x_gs + (int64_t)c0 * Tin); \
        {                                                                                                           \
            const int sg = lane >> seg_shift, t = t0 + (lane & (seg - 1));                                          \
            const unsigned vo = ((b0 + sg < B) && (t < T)) ? (unsigned)(((int64_t)sg * gy_bs + t) * 4) : NEF_OOB;    \
            _Pragma("unroll") for (int rr = 0; rr < GR; ++rr)                                                       \
                greg[rr] = nef_buf_f32(grs, vo, (unsigned)((wave_u + 4 * rr) * T * 4));                             \
        }                                                                                                           \
        {   /* main 64 positions of every row: uniform row offsets */                                               \
            const int sg = lane / segw;                                                                             \
            const int t = t0 + (lane - sg * segw) - PAD;                                                            \
            const bool ok = (b0 + sg < B) && (t >= 0) && (t < T);                                                   \
            NEF_BW_TAPS(t, i0_, i1_, l_)                                                                            \
            xk_m = ok;                                                                                              \
            lam_m = l_;                                                                                             \
            const unsigned vo0 = ok ? (unsigned)(((int64_t)sg * x_bs + i0_) * 4) : NEF_OOB;                         \
            const unsigned vo1 = ok ? (unsigned)(((int64_t)sg * x_bs + i1_) * 4) : NEF_OOB;                         \
            _Pragma("unroll") for (int rr = 0; rr < XRW; ++rr) {                                                    \
                xreg[rr][0] = nef_buf_f32(xrs, vo0, (unsigned)((wave_u + 4 * rr) * Tin * 4));                       \
                if constexpr (UP) xreg[rr][NS - 1] = nef_buf_f32(xrs, vo1, (unsigned)((wave_u + 4 * rr) * Tin * 4)); \
            }                                                                                                       \
            if (in_scale) {                                                                                         \
                const int64_t so = ok ? (int64_t)(b0 + sg) * sc_bs + (int64_t)g * sc_gs + c0 : 0;                   \
                _Pragma("unroll") for (int rr = 0; rr < XRW; ++rr) xreg[rr][0] *= in_scale[so + (ok ? wave + 4 * rr : 0)]; \
            }                                                                                                       \
        }                                                                                                           \
        _Pragma("unroll") for (int h = 0; h < NH; ++h) {   /* the few positions beyond 64: (row, e) per lane */      \
            const int idx = (int)threadIdx.x + 256 * h;                                                             \
            const int row = nh > 0 ? idx / nh : 0;                                                                  \
            const int r = 64 + idx - row * nh;                                                                      \
            const int sg = r / segw;                                                                                \
            const int t = t0 + (r - sg * segw) - PAD;                                                               \
            const bool ok = (nh > 0) && (row < CIT) && (b0 + sg < B) && (t >= 0) && (t < T);                        \
            NEF_BW_TAPS(t, i0_, i1_, l_)                                                                            \
            xk_h[h] = ok;                                                                                           \
            lam_h[h] = l_;                                                                                          \
            xh[h][0] = nef_buf_f32(xrs, ok ? (unsigned)(((int64_t)sg * x_bs + (int64_t)row * Tin + i0_) * 4) : NEF_OOB, 0); \
            if constexpr (UP)                                                                                       \
                xh[h][NS - 1] = nef_buf_f32(xrs, ok ? (unsigned)(((int64_t)sg * x_bs + (int64_t)row * Tin + i1_) * 4) : NEF_OOB, 0); \
            if (in_scale && ok) xh[h][0] *= in_scale[(int64_t)(b0 + sg) * sc_bs + (int64_t)g * sc_gs + c0 + row];   \
        }                                                                                                           \
    }
    // The Winograd forms run on whole 64-column tiles of ONE sample (nseg == 1), which allows a much shorter issue phase --
    // with only 48..56 matrix instructions per wave between two barriers (F(3,4) / F(4,4)) the ~200 instructions the generic
    // macro spends per tile were 17..19 % of the kernel (timing-only builds, profiles/r03_bwd_weight_ablation.md):
    //   * tile -> (sample, column) coordinates advance incrementally (no division per tile);
    //   * the gY tile comes in as 16-byte vectors -- a lane fetches 4 consecutive columns of one row, an instruction
    //     covers 4 rows: GR / 4 loads per wave instead of GR, and as many 16-byte LDS stores (two aligned 8-byte halves:
    //     rows are 8-byte aligned because T is even).  The last tile of a sample, whose columns end inside a vector,
    //     takes the dword path (reading past the row end would leave the tensor on its very last row);
    //   * the K - 1 halo positions per X row are addressed with compile-time divisors.
    constexpr bool FASTW = WINO != 0 && !UP;
    bool g_vec = false;                  // layout of greg: [q][4] vectors (rows wave*GR + 4q + lane/16) or one row per entry
    int nb0 = 0, ntq = 0, s_div = 0, s_mod = 0;
    if constexpr (FASTW) {
        nb0 = split / tps;
        ntq = split - nb0 * tps;
        s_div = S / tps;
        s_mod = S - s_div * tps;
    }
    const unsigned gq_vo = (unsigned)(((lane >> 4) * T + 4 * (lane & 15)) * 4);
#define NEF_BW_ISSUE_W()                                                                                              \
    {                                                                                                               \
        const int b0 = nb0, t0 = ntq * WT;                                                                          \
        pass_ld = AFF ? b0 / pro_Bp : 0;                                                                            \
        const __amdgpu_buffer_rsrc_t grs = nef_rsrc(gy + (int64_t)b0 * gy_bs + (int64_t)g * gy_gs + (int64_t)m0 * T); \
        const __amdgpu_buffer_rsrc_t xrs = nef_rsrc(x + (int64_t)b0 * x_bs + (int64_t)g * x_gs + (int64_t)c0 * Tin); \
        g_vec = t0 + WT <= T;                                                                                       \
        if (g_vec) {                                                                                                \
            _Pragma("unroll") for (int q = 0; q < GR / 4; ++q) {                                                    \
                const f32x4 v4_ = nef_buf_f32x4(grs, gq_vo, (unsigned)(((wave_u * GR + 4 * q) * T + t0) * 4));       \
                greg[4 * q] = v4_[0];                                                                               \
                greg[4 * q + 1] = v4_[1];                                                                           \
                greg[4 * q + 2] = v4_[2];                                                                           \
                greg[4 * q + 3] = v4_[3];                                                                           \
            }                                                                                                       \
        } else {                                                                                                    \
            const unsigned vo = (t0 + lane < T) ? (unsigned)((t0 + lane) * 4) : NEF_OOB;                            \
            _Pragma("unroll") for (int rr = 0; rr < GR; ++rr)                                                       \
                greg[rr] = nef_buf_f32(grs, vo, (unsigned)((wave_u + 4 * rr) * T * 4));                             \
        }                                                                                                           \
        {                                                                                                           \
            const int t = t0 + lane - PAD;                                                                          \
            const bool ok = (t >= 0) && (t < T);                                                                    \
            xk_m = ok;                                                                                              \
            const unsigned vo0 = ok ? (unsigned)(t * 4) : NEF_OOB;                                                  \
            _Pragma("unroll") for (int rr = 0; rr < XRW; ++rr)                                                      \
                xreg[rr][0] = nef_buf_f32(xrs, vo0, (unsigned)((wave_u + 4 * rr) * Tin * 4));                       \
            if (in_scale) {                                                                                         \
                const int64_t so = ok ? (int64_t)b0 * sc_bs + (int64_t)g * sc_gs + c0 : 0;                          \
                _Pragma("unroll") for (int rr = 0; rr < XRW; ++rr) xreg[rr][0] *= in_scale[so + (ok ? wave + 4 * rr : 0)]; \
            }                                                                                                       \
        }                                                                                                           \
        _Pragma("unroll") for (int h = 0; h < NH; ++h) {                                                            \
            const int idx = (int)threadIdx.x + 256 * h;                                                             \
            const int row = idx / (K > 1 ? K - 1 : 1);                                                              \
            const int t = t0 + 64 + (idx - row * (K - 1)) - PAD;                                                    \
            const bool ok = (row < CIT) && (t < T);                                                                 \
            xk_h[h] = ok;                                                                                           \
            xh[h][0] = nef_buf_f32(xrs, ok ? (unsigned)((row * Tin + t) * 4) : NEF_OOB, 0);                         \
            if (in_scale && ok) xh[h][0] *= in_scale[(int64_t)b0 * sc_bs + (int64_t)g * sc_gs + c0 + row];          \
        }                                                                                                           \
        nb0 += s_div;                                                                                               \
        ntq += s_mod;                                                                                               \
        if (ntq >= tps) {                                                                                           \
            ntq -= tps;                                                                                             \
            ++nb0;                                                                                                  \
        }                                                                                                           \
    }
    if (split < n_tiles) {
        if constexpr (FASTW) NEF_BW_ISSUE_W() else NEF_BW_ISSUE(split)
    }
#ifdef NEF_BW_SETPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
    for (int tile = split; tile < n_tiles; tile += S) {
        __syncthreads();
        if (!(NEF_ABL & 2) || tile == split) {
        if (FASTW && g_vec) {
            float* gp = GYl + (wave * GR + (lane >> 4)) * GYS + 4 * (lane & 15);
#pragma unroll
            for (int q = 0; q < GR / 4; ++q) {
                f32x2 lo2, hi2;
                lo2[0] = greg[4 * q];
                lo2[1] = greg[4 * q + 1];
                hi2[0] = greg[4 * q + 2];
                hi2[1] = greg[4 * q + 3];
                *reinterpret_cast<f32x2*>(gp + 4 * q * GYS) = lo2;
                *reinterpret_cast<f32x2*>(gp + 4 * q * GYS + 2) = hi2;
            }
        } else {
#pragma unroll
            for (int rr = 0; rr < GR; ++rr) GYl[(wave + 4 * rr) * GYS + lane] = greg[rr];
        }
        {
            const int prow0 = AFF ? pass_ld * G * Cig + g * Cig + c0 : 0;
#pragma unroll
            for (int rr = 0; rr < XRW; ++rr) {
                float v = xreg[rr][0];
                if constexpr (PRO != 0) {
                    float pa = 1.f, pb = 0.f;
                    if constexpr (AFF) {
                        pa = pro_a[prow0 + wave_u + 4 * rr];
                        pb = pro_b[prow0 + wave_u + 4 * rr];
                        v = fmaxf(fmaf(v, pa, pb), 0.f);
                    }
                    if constexpr (UP) {
                        float v1 = xreg[rr][NS - 1];
                        if constexpr (AFF) v1 = fmaxf(fmaf(v1, pa, pb), 0.f);
                        v = (1.f - lam_m) * v + lam_m * v1;
                    }
                    v = xk_m ? v : 0.f;
                }
                Xl[(wave + 4 * rr) * XS + lane] = v;
            }
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const int idx = (int)threadIdx.x + 256 * h;
                const int row = FASTW ? idx / (K > 1 ? K - 1 : 1) : (nh > 0 ? idx / nh : 0);
                float v = xh[h][0];
                if constexpr (PRO != 0) {
                    float pa = 1.f, pb = 0.f;
                    if constexpr (AFF) {
                        if (xk_h[h]) {
                            pa = pro_a[prow0 + row];
                            pb = pro_b[prow0 + row];
                        }
                        v = fmaxf(fmaf(v, pa, pb), 0.f);
                    }
                    if constexpr (UP) {
                        float v1 = xh[h][NS - 1];
                        if constexpr (AFF) v1 = fmaxf(fmaf(v1, pa, pb), 0.f);
                        v = (1.f - lam_h[h]) * v + lam_h[h] * v1;
                    }
                    v = xk_h[h] ? v : 0.f;
                }
                if (nh > 0 && row < CIT) Xl[row * XS + 64 + idx - row * nh] = v;
            }
        }
        }
        __syncthreads();
        if constexpr (!(NEF_ABL & 1)) if (tile + S < n_tiles) {
            if constexpr (FASTW) NEF_BW_ISSUE_W() else NEF_BW_ISSUE(tile + S)
        }
        if constexpr (WINO == 4 || WINO == 5) {
            constexpr int NSTEP = WT / 8;
            const float* ga = GYl + (wco * 32 + lo) * GYS + 4 * hi;
            const float* xb = Xl + ((wci * TCI) * 32 + lo) * XS + 4 * hi;          // position p of a staged row holds x[t0 + p - 3]
            if constexpr (WINO == 4) {
                // taps 0..3: quad j, gy[4j..4j+3] against d_m = x[4j-3+m], m = 0..6 (four aligned 8-byte words)
                f32x2 fg[2][2], fx[2][4];
#define NEF_BW7A_LOAD(S_, BUF)                                                                                       \
    {                                                                                                               \
        fg[BUF][0] = *reinterpret_cast<const f32x2*>(ga + 8 * (S_));                                                \
        fg[BUF][1] = *reinterpret_cast<const f32x2*>(ga + 8 * (S_) + 2);                                            \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_)                                                            \
            fx[BUF][q_] = *reinterpret_cast<const f32x2*>(xb + 8 * (S_) + 2 * q_);                                  \
    }
                NEF_BW7A_LOAD(0, 0)
#pragma unroll
                for (int s_ = 0; s_ < NSTEP; ++s_) {
                    if ((NEF_ABL & 4) ? s_ == 0 : s_ + 1 < NSTEP) NEF_BW7A_LOAD(s_ + 1, (s_ + 1) & 1)
                    const float g0 = fg[s_ & 1][0][0], g1 = fg[s_ & 1][0][1], g2 = fg[s_ & 1][1][0], g3 = fg[s_ & 1][1][1];
                    const f32x2* d = fx[s_ & 1];
                    const float d0 = d[0][0], d1 = d[0][1], d2 = d[1][0], d3 = d[1][1], d4 = d[2][0], d5 = d[2][1], d6 = d[3][0];
                    float u[7], v[7];
                    {
                        const float e02 = g0 + g2, e13 = g1 + g3;
                        const float f02 = fmaf(4.f, g2, g0), f13 = 2.f * fmaf(4.f, g3, g1);
                        u[0] = g0;
                        u[1] = e02 + e13;
                        u[2] = e02 - e13;
                        u[3] = f02 + f13;
                        u[4] = f02 - f13;
                        u[5] = fmaf(0.125f, g3, fmaf(0.25f, g2, fmaf(0.5f, g1, g0)));      // point 1/2
                        u[6] = g3;                                                          // infinity
                    }
                    v[5] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
                    v[0] = fmaf(-2.f, d0, fmaf(2.5f, d2, fmaf(-0.5f, d4, v[5])));
                    v[6] = fmaf(-2.f, d1, fmaf(2.5f, d3, fmaf(-0.5f, d5, fmaf(4.f, d2, fmaf(-5.f, d4, d6)))));
                    {
                        const float p = fmaf(2.f, d2, fmaf(-4.f, d3, fmaf(-0.5f, d4, d5)));
                        const float q = fmaf(2.f, d1, fmaf(-4.f, d2, fmaf(-0.5f, d3, d4)));
                        v[1] = p + q;
                        v[2] = p - q;
                        const float d24 = d2 - d4;
                        const float p2 = fmaf(0.5f, d24, d5 - d3), q2 = fmaf(-2.f, d24, d1 - d3);
                        v[3] = p2 + q2;
                        v[4] = p2 - q2;
                    }
#pragma unroll
                    for (int n = 0; n < 7; ++n)
                        acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[n], v[n], acc[0][n], 0, 0, 0);
                }
#undef NEF_BW7A_LOAD
            } else {
                // taps 4..6: the transposed F(3,4) of the K = 3 gradients on the window x[4j+1 .. 4j+6] (three aligned words)
                f32x2 fg[2][2], fx[2][3];
#define NEF_BW7B_LOAD(S_, BUF)                                                                                       \
    {                                                                                                               \
        fg[BUF][0] = *reinterpret_cast<const f32x2*>(ga + 8 * (S_));                                                \
        fg[BUF][1] = *reinterpret_cast<const f32x2*>(ga + 8 * (S_) + 2);                                            \
        _Pragma("unroll") for (int q_ = 0; q_ < 3; ++q_)                                                            \
            fx[BUF][q_] = *reinterpret_cast<const f32x2*>(xb + 8 * (S_) + 4 + 2 * q_);                              \
    }
                NEF_BW7B_LOAD(0, 0)
#pragma unroll
                for (int s_ = 0; s_ < NSTEP; ++s_) {
                    if ((NEF_ABL & 4) ? s_ == 0 : s_ + 1 < NSTEP) NEF_BW7B_LOAD(s_ + 1, (s_ + 1) & 1)
                    const float g0 = fg[s_ & 1][0][0], g1 = fg[s_ & 1][0][1], g2 = fg[s_ & 1][1][0], g3 = fg[s_ & 1][1][1];
                    float u[6];
                    {
                        const float e02 = g0 + g2, e13 = g1 + g3;
                        const float f02 = fmaf(4.f, g2, g0), f13 = 2.f * fmaf(4.f, g3, g1);
                        u[0] = g0;
                        u[1] = e02 + e13;
                        u[2] = e02 - e13;
                        u[3] = f02 + f13;
                        u[4] = f02 - f13;
                        u[5] = g3;
                    }
                    const f32x2* d = fx[s_ & 1];
                    const float d0 = d[0][0], d1 = d[0][1], d2 = d[1][0], d3 = d[1][1], d4 = d[2][0], d5 = d[2][1];
                    float v[6];
                    const float t1 = fmaf(-4.f, d2, d4), t2 = fmaf(-4.f, d1, d3);
                    const float t3 = d4 - d2, t4 = 2.f * (d3 - d1);
                    v[0] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
                    v[1] = t1 + t2;
                    v[2] = t1 - t2;
                    v[3] = t3 + t4;
                    v[4] = t3 - t4;
                    v[5] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
#pragma unroll
                    for (int n = 0; n < 6; ++n)
                        acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[n], v[n], acc[0][n], 0, 0, 0);
                }
#undef NEF_BW7B_LOAD
            }
        } else if constexpr (WINO == 2) {
            // transposed F(3,4): 8 reduction steps per 64-column tile, each over two output QUADS (MFMA k = quad): a lane
            // reads its gY row's quad (two aligned 8-byte words) and its X row's six inputs x[4j-1 .. 4j+4] (three), one
            // step ahead of use; 6 MFMAs per 8 columns where F(3,2) issues 8 and the direct form 12
            constexpr int NSTEP = WT / 8;
            const float* ga = GYl + (wco * 32 + lo) * GYS + 4 * hi;
            const float* xb = Xl + ((wci * TCI) * 32 + lo) * XS + 4 * hi;
            f32x2 fg[2][2], fx[2][TCI][3];
#define NEF_BW4_LOAD(S_, BUF)                                                                                        \
    {                                                                                                               \
        fg[BUF][0] = *reinterpret_cast<const f32x2*>(ga + 8 * (S_));                                                \
        fg[BUF][1] = *reinterpret_cast<const f32x2*>(ga + 8 * (S_) + 2);                                            \
        _Pragma("unroll") for (int i = 0; i < TCI; ++i)                                                             \
            _Pragma("unroll") for (int q_ = 0; q_ < 3; ++q_)                                                        \
                fx[BUF][i][q_] = *reinterpret_cast<const f32x2*>(xb + i * 32 * XS + 8 * (S_) + 2 * q_);             \
    }
            NEF_BW4_LOAD(0, 0)
#pragma unroll
            for (int s_ = 0; s_ < NSTEP; ++s_) {
                if ((NEF_ABL & 4) ? s_ == 0 : s_ + 1 < NSTEP) NEF_BW4_LOAD(s_ + 1, (s_ + 1) & 1)
                const float g0 = fg[s_ & 1][0][0], g1 = fg[s_ & 1][0][1], g2 = fg[s_ & 1][1][0], g3 = fg[s_ & 1][1][1];
                float u[6];
                {
                    const float e02 = g0 + g2, e13 = g1 + g3;
                    const float f02 = fmaf(4.f, g2, g0), f13 = 2.f * fmaf(4.f, g3, g1);
                    u[0] = g0;
                    u[1] = e02 + e13;
                    u[2] = e02 - e13;
                    u[3] = f02 + f13;
                    u[4] = f02 - f13;
                    u[5] = g3;
                }
#pragma unroll
                for (int i = 0; i < TCI; ++i) {
                    const f32x2* d = fx[s_ & 1][i];
                    const float d0 = d[0][0], d1 = d[0][1], d2 = d[1][0], d3 = d[1][1], d4 = d[2][0], d5 = d[2][1];
                    float v[6];
                    const float t1 = fmaf(-4.f, d2, d4), t2 = fmaf(-4.f, d1, d3);
                    const float t3 = d4 - d2, t4 = 2.f * (d3 - d1);
                    v[0] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
                    v[1] = t1 + t2;
                    v[2] = t1 - t2;
                    v[3] = t3 + t4;
                    v[4] = t3 - t4;
                    v[5] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
#pragma unroll
                    for (int n = 0; n < 6; ++n)
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[n], v[n], acc[i][n], 0, 0, 0);
                }
            }
#undef NEF_BW4_LOAD
        } else if constexpr (WINO) {
            // 16 reduction steps per 64-column tile, each over two output pairs (MFMA k = pair): a lane reads its gY row's
            // pair and its X row's 4 (K = 3) or 8 (K = 7) inputs as aligned 8-byte words, one step ahead of use
            constexpr int NXV = K == 3 ? 2 : 4;
            constexpr int NSTEP = WT / 4;
            const float* ga = GYl + (wco * 32 + lo) * GYS + 2 * hi;
            const float* xb = Xl + ((wci * TCI) * 32 + lo) * XS + 2 * hi;
            f32x2 fg[2], fx[2][TCI][NXV];
#define NEF_BWW_LOAD(S_, BUF)                                                                                        \
    {                                                                                                               \
        fg[BUF] = *reinterpret_cast<const f32x2*>(ga + 4 * (S_));                                                   \
        _Pragma("unroll") for (int i = 0; i < TCI; ++i)                                                             \
            _Pragma("unroll") for (int q_ = 0; q_ < NXV; ++q_)                                                      \
                fx[BUF][i][q_] = *reinterpret_cast<const f32x2*>(xb + i * 32 * XS + 4 * (S_) + 2 * q_);             \
    }
            NEF_BWW_LOAD(0, 0)
#pragma unroll
            for (int s_ = 0; s_ < NSTEP; ++s_) {
                if ((NEF_ABL & 4) ? s_ == 0 : s_ + 1 < NSTEP) NEF_BWW_LOAD(s_ + 1, (s_ + 1) & 1)
                const float g0 = fg[s_ & 1][0], g1 = fg[s_ & 1][1];
                float u[4];
                u[0] = g0;
                u[1] = g0 + g1;
                u[2] = g0 - g1;
                u[3] = g1;
#pragma unroll
                for (int i = 0; i < TCI; ++i) {
                    const f32x2* d = fx[s_ & 1][i];
                    if constexpr (WINO == 3) {
                        // K = 7 split 4 + 3: taps 0..3 through the transposed F(4,2) (points 0, 1, -1, 2, inf: 5 products
                        // per pair for 4 taps), taps 4..6 through F(3,2) on x[4..7]: 9 MFMAs per 4 columns instead of 10
                        const float x0 = d[0][0], x1 = d[0][1], x2 = d[1][0], x3 = d[1][1], x4 = d[2][0], x5 = d[2][1],
                                    x6 = d[3][0], x7 = d[3][1];
                        float va[5], vb[4];
                        const float p13 = x3 - x1;
                        va[0] = fmaf(2.f, x0 - x2, p13);                 // 2x0 - x1 - 2x2 + x3
                        va[1] = fmaf(-2.f, x1, x3 - x2);                 // -2x1 - x2 + x3
                        va[2] = fmaf(2.f, x1, fmaf(-3.f, x2, x3));       // 2x1 - 3x2 + x3
                        va[3] = p13;                                     // -x1 + x3
                        va[4] = fmaf(2.f, x1 - x3, x4 - x2);             // 2x1 - x2 - 2x3 + x4
                        const float ua3 = fmaf(2.f, g1, g0);
                        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[0], va[0], acc[i][0], 0, 0, 0);
                        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[1], va[1], acc[i][1], 0, 0, 0);
                        acc[i][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[2], va[2], acc[i][2], 0, 0, 0);
                        acc[i][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua3, va[3], acc[i][3], 0, 0, 0);
                        acc[i][4] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[3], va[4], acc[i][4], 0, 0, 0);
                        vb[0] = x4 - x6;
                        vb[1] = x5 + x6;
                        vb[2] = x6 - x5;
                        vb[3] = x5 - x7;
#pragma unroll
                        for (int n = 0; n < 4; ++n)
                            acc[i][5 + n] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[n], vb[n], acc[i][5 + n], 0, 0, 0);
                        continue;
                    }
                    float v[4];
                    v[0] = d[0][0] - d[1][0];
                    v[1] = d[0][1] + d[1][0];
                    v[2] = d[1][0] - d[0][1];
                    v[3] = d[0][1] - d[1][1];
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[n], v[n], acc[i][n], 0, 0, 0);
                    if constexpr (K == 7) {
                        float w[4];
                        w[0] = d[1][1] - d[2][1];
                        w[1] = d[2][0] + d[2][1];
                        w[2] = d[2][1] - d[2][0];
                        w[3] = d[2][0] - d[3][0];
#pragma unroll
                        for (int n = 0; n < 4; ++n)
                            acc[i][4 + n] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[n], w[n], acc[i][4 + n], 0, 0, 0);
                        acc[i][8] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, d[3][0], acc[i][8], 0, 0, 0);
                        acc[i][8] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, d[3][1], acc[i][8], 0, 0, 0);
                    }
                }
            }
#undef NEF_BWW_LOAD
        } else
        // 32 reduction steps (2 columns each) per 64-column tile, software-pipelined like the forward kernel: the
        // fragments of step group gi+1 are read from LDS while the MFMAs of group gi issue.  Column 2*step of the
        // gY tile is linear; in the X tile every sample segment carries K-1 halo columns.
        {
            constexpr int GS = 2;
            constexpr int NG = (WT / 2) / GS;
            const float* ga = GYl + (wco * 32 + lo) * GYS + hi;
            const float* xb = Xl + ((wci * TCI) * 32 + lo) * XS + hi;
            float fa[2][GS], fb[2][GS][TCI][K];
#define NEF_BW_LOAD(GI, BUF)                                                                                         \
    _Pragma("unroll") for (int s_ = 0; s_ < GS; ++s_) {                                                             \
        const int col_ = 2 * ((GI) * GS + s_);                                                                      \
        const int xo_ = col_ + (col_ >> seg_shift) * (K - 1);                                                       \
        fa[BUF][s_] = ga[col_];                                                                                     \
        _Pragma("unroll") for (int i = 0; i < TCI; ++i)                                                             \
            _Pragma("unroll") for (int k = 0; k < K; ++k) fb[BUF][s_][i][k] = xb[i * 32 * XS + xo_ + k];            \
    }
            NEF_BW_LOAD(0, 0)
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                if (gi + 1 < NG) NEF_BW_LOAD(gi + 1, (gi + 1) & 1)
#pragma unroll
                for (int s_ = 0; s_ < GS; ++s_)
#pragma unroll
                    for (int i = 0; i < TCI; ++i)
#pragma unroll
                        for (int k = 0; k < K; ++k)
                            acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[gi & 1][s_], fb[gi & 1][s_][i][k],
                                                                             acc[i][k], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < GS * TCI * K; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
#undef NEF_BW_LOAD
        }
    }
#undef NEF_BW_ISSUE
#undef NEF_BW_ISSUE_W
#undef NEF_BW_TAPS
    if constexpr ((NEF_ABL & 8) != 0) if (T >= 0) return;      // run-time true: the stores below are dead at run time only
    // partials: ws[split][g][k][co][ci]
#pragma unroll
    for (int i = 0; i < TCI; ++i) {
        const int ci = c0 + (wci * TCI + i) * 32 + lo;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if constexpr (WINO == 4 || WINO == 5) if ((k < 4) != (WINO == 4)) continue;      // a launch owns its tap group only
            float* dst = ws + ((((int64_t)split * G + g) * K + k) * Cog) * Cig;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                float v;
                if constexpr (WINO == 4 || WINO == 5) {
                    const float m1 = acc[i][1][r], m2 = acc[i][2][r], m3 = acc[i][3][r], m4 = acc[i][4][r], m5 = acc[i][5][r];
                    if (k < 4) {        // gW = G^T M with the F(4,4) filter-transform matrix (points 0, 1, -1, 2, -2, 1/2, inf)
                        v = (k == 0) ? fmaf(-0.5f, acc[i][0][r], fmaf(m5, 32.f / 45.f, fmaf(m3, 1.f / 36.f, fmaf(m4, -1.f / 60.f, (m2 * (1.f / 9.f) - m1 * (1.f / 3.f))))))
                          : (k == 1) ? fmaf(m5, 16.f / 45.f, fmaf(m3, 1.f / 18.f, fmaf(m4, 1.f / 30.f, -(m2 * (1.f / 9.f) + m1 * (1.f / 3.f)))))
                          : (k == 2) ? fmaf(m5, 8.f / 45.f, fmaf(m3, 1.f / 9.f, fmaf(m4, -1.f / 15.f, (m2 * (1.f / 9.f) - m1 * (1.f / 3.f)))))
                                     : fmaf(m5, 4.f / 45.f, fmaf(m3, 2.f / 9.f, fmaf(m4, 2.f / 15.f, -(m2 * (1.f / 9.f) + m1 * (1.f / 3.f))))) + acc[i][NACC - 1][r];
                    } else {            // taps 4..6: G^T of F(4,3) on tiles 0..5, as WINO = 2
                        const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                        v = (k == 4) ? fmaf(0.25f, acc[i][0][r], fmaf(s34, 1.f / 24.f, -s12 * (1.f / 6.f)))
                          : (k == 5) ? fmaf(d34, 1.f / 12.f, -d12 * (1.f / 6.f))
                                     : (s34 - s12) * (1.f / 6.f) + m5;
                    }
                } else if constexpr (WINO == 2) {      // gW = G^T M with the F(4,3) filter-transform matrix G
                    const float s12 = acc[i][1][r] + acc[i][2][r], d12 = acc[i][1][r] - acc[i][2][r];
                    const float s34 = acc[i][3][r] + acc[i][4][r], d34 = acc[i][3][r] - acc[i][4][r];
                    v = (k == 0) ? fmaf(0.25f, acc[i][0][r], fmaf(s34, 1.f / 24.f, -s12 * (1.f / 6.f)))
                      : (k == 1) ? fmaf(d34, 1.f / 12.f, -d12 * (1.f / 6.f))
                                 : (s34 - s12) * (1.f / 6.f) + acc[i][5][r];
                } else if constexpr (WINO == 3) {
                    if (k < 4) {        // gW = G^T M with F(2,4)'s filter-transform matrix G (points 0, 1, -1, 2, inf)
                        const float m0 = acc[i][0][r], m1 = acc[i][1][r], m2 = acc[i][2][r], m3 = acc[i][3][r];
                        v = (k == 0) ? 0.5f * (m0 - m1) + (m3 - m2) * (1.f / 6.f)
                          : (k == 1) ? fmaf(m2, 1.f / 6.f, fmaf(m3, 1.f / 3.f, -0.5f * m1))
                          : (k == 2) ? fmaf(m3, 2.f / 3.f, fmaf(m2, -1.f / 6.f, -0.5f * m1))
                                     : fmaf(m3, 4.f / 3.f, fmaf(m2, 1.f / 6.f, -0.5f * m1)) + acc[i][4][r];
                    } else {            // taps 4..6: the F(3,2) output transform on tiles 5..8
                        const float hs = 0.5f * (acc[i][6][r] + acc[i][7][r]);
                        v = (k == 4) ? acc[i][5][r] + hs : (k == 5) ? 0.5f * (acc[i][6][r] - acc[i][7][r]) : hs - acc[i][8][r];
                    }
                } else if constexpr (WINO) {
                    if (k == 6) {
                        v = acc[i][NACC - 1][r];
                    } else {
                        const int q = 4 * (k / 3);              // transform group of this tap
                        const float hs = 0.5f * (acc[i][q + 1][r] + acc[i][q + 2][r]);
                        v = (k % 3 == 0) ? acc[i][q][r] + hs
                          : (k % 3 == 1) ? 0.5f * (acc[i][q + 1][r] - acc[i][q + 2][r])
                                         : hs - acc[i][q + 3][r];
                    }
                } else {
                    v = acc[i][k < NACC ? k : 0][r];
                }
                dst[(int64_t)co * Cig + ci] = v;
            }
        }
    }
}

// gw[g*Cog+co][ci][k] = sum_split ws[split][g][k][co][ci]
__global__ void conv_bwd_weight_reduce(const float* __restrict__ ws, float* __restrict__ gw, int G, int Cog, int Cig,
                                       int K, int S) {
    const int64_t n = (int64_t)G * K * Cog * Cig;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cig);
        int64_t r = i / Cig;
        const int co = (int)(r % Cog);
        r /= Cog;
        const int k = (int)(r % K);
        const int g = (int)(r / K);
        // fixed summation order (deterministic); loads issued eight at a time so they overlap
        float s = 0.f;
        int sp = 0;
        for (; sp + 8 <= S; sp += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ws[(int64_t)(sp + u) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; sp < S; ++sp) s += ws[(int64_t)sp * n + i];
        gw[(((int64_t)g * Cog + co) * Cig + ci) * K + k] = s;
    }
}

struct BwdWeightPlan {
    int wco, tci, m_tiles, ci_chunks, S;
    ColTiling ct;
};

static bool plan_bwd_weight(int B, int T, int G, int Cig, int Cog, int K, BwdWeightPlan* p, int pro_mode = 0,
                            int wino = 0) {
    if (!(K == 1 || K == 3 || K == 7)) return false;
    if (Cog % 128 == 0) p->wco = 4;
    else if (Cog % 64 == 0) p->wco = 2;
    else return false;
    const int wci = 4 / p->wco;
    p->tci = (K <= 3 && Cig % (64 * wci) == 0) ? 2 : 1;
    if (pro_mode != 0 && p->wco == 2) p->tci = 1;      // keep the doubled staging registers within budget
    if (wino && p->wco == 2) p->tci = 1;               // 4 accumulator tiles per ci tile: same budget
    if (wino == 2) p->tci = 1;                         // F(3,4): 6 accumulator tiles per ci tile
    if (wino == 4 && K != 7) return false;             // K = 7 tap split across two launches (7 resp. 6 tiles per wave)
    const int cit = 32 * p->tci * (4 / p->wco);
    if (Cig % cit != 0) return false;
    p->m_tiles = Cog / (32 * p->wco);
    p->ci_chunks = Cig / cit;
    p->ct = make_tiling(B, T, WT);
    const int base = G * p->m_tiles * p->ci_chunks;
#ifndef NEF_BW_TARGET
#define NEF_BW_TARGET 768
#endif
    int S = (NEF_BW_TARGET + base - 1) / base;
    if (S > p->ct.n_tiles) S = p->ct.n_tiles;
    if (S < 1) S = 1;
    p->S = S;
    return true;
}

template <int K, int WCO, int TCI, int PRO = 0, int WINO = 0>
static int launch_bwd_weight(BwdWeightPlan& p, const float* x, int64_t x_bs, int64_t x_gs, const float* in_scale,
                             int64_t sc_bs, int64_t sc_gs, const float* gy, int64_t gy_bs, int64_t gy_gs, float* ws,
                             int B, int T, int G, int Cig, int Cog, hipStream_t st, const float* pro_a = nullptr,
                             const float* pro_b = nullptr, int pro_Bp = 1, int fixed_S = 0) {
    constexpr int WCI = 4 / WCO;
    constexpr int MT = 32 * WCO;
    constexpr int CIT = 32 * TCI * WCI;
    constexpr int GYS = WINO ? 66 : WT + 1;
    constexpr int XS = WINO ? (K == 3 ? 66 : 98) : ((WT + (WT / 16) * (K - 1)) | 1);
    constexpr size_t lds = (size_t)(MT * GYS + CIT * XS) * sizeof(float);
    static unsigned long long lds_set = 0;      // per-device bits, see nef_ensure_dyn_lds
    if (int e = nef_ensure_dyn_lds(reinterpret_cast<const void*>(&conv_bwd_weight_kernel<K, WCO, TCI, PRO, WINO>), lds, &lds_set)) return e;
    // Split count: the workgroups are persistent over their share of the column tiles, so the launch should be exactly
    // ONE round of resident workgroups (occupancy x CUs) -- a partial second round leaves half the chip idle for a whole
    // workgroup lifetime (measured: 768 workgroups on 512 slots cost 5..8 % against 504..512).  Never more splits than the
    // workspace was sized for (p.S from plan_bwd_weight is that bound).
    if (fixed_S > 0) {       // second half of a two-launch gradient: the split count of the first half, whatever fits
        p.S = fixed_S;
    } else {
        static int resident_dev[64] = {0};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        int resident = __atomic_load_n(&resident_dev[dev & 63], __ATOMIC_ACQUIRE);
        if (resident == 0) {
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(
                    &per_cu, reinterpret_cast<const void*>(&conv_bwd_weight_kernel<K, WCO, TCI, PRO, WINO>), 256, lds) !=
                hipSuccess || per_cu <= 0)
                per_cu = 2;
            resident = per_cu * nef_cu_count();
            __atomic_store_n(&resident_dev[dev & 63], resident, __ATOMIC_RELEASE);
        }
        const int base = G * p.m_tiles * p.ci_chunks;
        int S = resident / base;
        if (S > p.S) S = p.S;
        if (S < 1) S = 1;
        p.S = S;
    }
    const int64_t blocks = (int64_t)p.S * G * p.m_tiles * p.ci_chunks;
    hipLaunchKernelGGL((conv_bwd_weight_kernel<K, WCO, TCI, PRO, WINO>), dim3((unsigned)blocks), dim3(256), lds, st, x, x_bs,
                       x_gs, in_scale, sc_bs, sc_gs, gy, gy_bs, gy_gs, ws, B, T, G, Cig, Cog, p.ct.seg_shift, p.ct.nseg,
                       p.ct.tps, p.ct.n_tiles, p.m_tiles, p.ci_chunks, p.S, pro_a, pro_b, pro_Bp);
    return nef_launch_status();
}

__device__ __forceinline__ void pack_plain_elem(const float* __restrict__ w, float* __restrict__ wp, int G, int Cog,
                                                int Cig, int K, int flip, int64_t i) {
    // i indexes the PACKED tensor
    int64_t r = i;
    int co, ci;
    if (!flip) {   // [g][k][ci][co]
        co = (int)(r % Cog); r /= Cog;
        ci = (int)(r % Cig); r /= Cig;
    } else {       // [g][k][co][ci]
        ci = (int)(r % Cig); r /= Cig;
        co = (int)(r % Cog); r /= Cog;
    }
    const int k = (int)(r % K);
    const int g = (int)(r / K);
    const int ks = flip ? (K - 1 - k) : k;
    wp[i] = w[(((int64_t)g * Cog + co) * Cig + ci) * K + ks];
}

__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int G, int Cog, int Cig, int K,
                                   int flip) {
    const int64_t n = (int64_t)G * Cog * Cig * K;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        pack_plain_elem(w, wp, G, Cog, Cig, K, flip, i);
}

// Every operand of a forward (or backward) pass in ONE launch: blockIdx.y selects the descriptor, which travels by value
// in the kernel arguments (no device-side table to keep in sync).
constexpr int PACK_MULTI_MAX = 48;
struct PackTable { nef_pack_desc d[PACK_MULTI_MAX]; };

__global__ void pack_weights_multi_kernel(PackTable t) {
    const nef_pack_desc& d = t.d[blockIdx.y];
    const int64_t n = (int64_t)d.G * d.Cog * d.Cig * (d.wino ? 1 : d.K);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (d.wino == 2) pack_wino4_elem(d.w, d.wp, d.G, d.Cog, d.Cig, d.K, d.transpose_flip, i);
        else if (d.wino) pack_wino_elem(d.w, d.wp, d.G, d.Cog, d.Cig, d.K, d.transpose_flip, i);
        else pack_plain_elem(d.w, d.wp, d.G, d.Cog, d.Cig, d.K, d.transpose_flip, i);
    }
}

__global__ void chan_sum_partial(const float* __restrict__ x, double* __restrict__ part, int B, int C, int T,
                                 int nsplit) {
    __shared__ double sm[4];
    const int c = blockIdx.x % C;
    const int sp = blockIdx.x / C;
    double s = 0.0;
    const int nb = sp < B ? (B - sp + nsplit - 1) / nsplit : 0;
    if (T >= 512 && (T & 1) == 0) {
        // long rows (8-byte aligned: T even): a thread walks pairs of one row, four rows in flight -- no per-element
        // division, 8-byte loads, 2 KB per wave instruction (the flat index space below ran at 37 % of HBM)
        const int T2 = T >> 1;
        int bi = 0;
        for (; bi + 4 <= nb; bi += 4) {
            const float2* r0 = reinterpret_cast<const float2*>(x + ((int64_t)(sp + bi * nsplit) * C + c) * T);
            const int64_t rs = (int64_t)nsplit * C * T2;           // row-to-row distance in float2
            for (int t = threadIdx.x; t < T2; t += blockDim.x) {
                const float2 a0 = r0[t], a1 = r0[rs + t], a2 = r0[2 * rs + t], a3 = r0[3 * rs + t];
                s += ((double)a0.x + (double)a0.y) + ((double)a1.x + (double)a1.y) +
                     (((double)a2.x + (double)a2.y) + ((double)a3.x + (double)a3.y));
            }
        }
        for (; bi < nb; ++bi) {
            const float2* r0 = reinterpret_cast<const float2*>(x + ((int64_t)(sp + bi * nsplit) * C + c) * T);
            for (int t = threadIdx.x; t < T2; t += blockDim.x) {
                const float2 a0 = r0[t];
                s += (double)a0.x + (double)a0.y;
            }
        }
    } else {
        // the split's rows b = sp, sp + nsplit, .. as ONE index space (row, t): short rows (the 16/32-sample ROI latents, the
        // 6-sample z2 window) keep all 256 threads busy instead of T of them
        for (int i = threadIdx.x; i < nb * T; i += blockDim.x) {
            const int bi = i / T, t = i - bi * T;
            s += (double)x[((int64_t)(sp + bi * nsplit) * C + c) * T + t];
        }
    }
    s = nef_block_sum_d(s, sm);
    if (threadIdx.x == 0) part[(int64_t)sp * C + c] = s;
}

__global__ void chan_sum_final(const double* __restrict__ part, float* __restrict__ out, int C, int nsplit) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int sp = 0; sp < nsplit; ++sp) s += part[(int64_t)sp * C + c];
    out[c] = (float)s;
}

constexpr int CHAN_SUM_SPLIT = 16;

}  // namespace

// conv_h2.hip: direct convolutions on exact fp16 splits of the fp32 operands (conv args wino = 3)
__attribute__((visibility("hidden"))) bool nef_h2_ok(const nef_conv_args* a);
__attribute__((visibility("hidden"))) int nef_h2_launch(const nef_conv_args* a, hipStream_t st);
__attribute__((visibility("hidden"))) int nef_h2_pack(const nef_pack_desc* descs, int n, hipStream_t st);

__attribute__((visibility("hidden"))) int nef_mfma_wino4_fwd(const nef_conv_args* a, hipStream_t st);
__attribute__((visibility("hidden"))) int nef_mfma_wino_fwd(const nef_conv_args* a, hipStream_t st);
__attribute__((visibility("hidden"))) int nef_mfma_bww_wino4(const float* x, int64_t x_bs, int64_t x_gs, const float* in_scale, int64_t sc_bs,
                              int64_t sc_gs, const float* pro_a, const float* pro_b, int pro_mode, int pro_Bp,
                              const float* gy, int64_t gy_bs, int64_t gy_gs, float* gw, void* ws, size_t ws_bytes, int B,
                              int T, int G, int Cin_g, int Cout_g, int K, nef_stream_t stream);

#if NEF_PART(1)
extern "C" {

int nef_abi_version(void) { return 18; }

int nef_pack_weight(const float* w, float* wp, int G, int Cog, int Cig, int K, int transpose_flip,
                    nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(w && wp, NEF_E_NULL);
    NEF_REQUIRE(G > 0 && Cog > 0 && Cig > 0 && K > 0, NEF_E_SHAPE);
    const int64_t n = (int64_t)G * Cog * Cig * K;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(nef_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, w, wp, G,
                       Cog, Cig, K, transpose_flip);
    return nef_launch_status();
}

int nef_pack_weight_wino(const float* w, float* wp, int G, int Cog, int Cig, int K, int transpose_flip,
                         nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(w && wp, NEF_E_NULL);
    NEF_REQUIRE(G > 0 && Cog > 0 && Cig > 0 && (K == 3 || K == 7), NEF_E_SHAPE);
    const int64_t n = (int64_t)G * Cog * Cig;
    hipLaunchKernelGGL(pack_weight_wino_kernel, dim3(nef_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, w, wp,
                       G, Cog, Cig, K, transpose_flip);
    return nef_launch_status();
}

int nef_pack_weight_wino4(const float* w, float* wp, int G, int Cog, int Cig, int K, int transpose_flip,
                          nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(w && wp, NEF_E_NULL);
    NEF_REQUIRE(G > 0 && Cog > 0 && Cig > 0 && (K == 3 || K == 7), NEF_E_SHAPE);
    const int64_t n = (int64_t)G * Cog * Cig;
    hipLaunchKernelGGL(pack_weight_wino4_kernel, dim3(nef_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, w, wp,
                       G, Cog, Cig, K, transpose_flip);
    return nef_launch_status();
}

int nef_pack_weight_h2(const float* w, void* wp, int G, int Cog, int Cig, int K, int transpose_flip, nef_stream_t stream) {
    NEF_ENTER();
    nef_pack_desc d;
    d.w = w;
    d.wp = (float*)wp;
    d.G = G, d.Cog = Cog, d.Cig = Cig, d.K = K, d.transpose_flip = transpose_flip, d.wino = 3;
    d.src_mode = 0, d.src_Cr = 0;
    return nef_h2_pack(&d, 1, (hipStream_t)stream);
}

size_t nef_pack_weight_h2_bytes(int G, int Cog, int Cig, int K, int transpose_flip) {
    if (G <= 0 || Cog <= 0 || Cig <= 0 || K <= 0) return 0;
    return ((size_t)G * K * Cog * Cig + (size_t)G * (transpose_flip ? Cig : Cog)) * sizeof(float);
}

int nef_pack_weights(const nef_pack_desc* descs, int n, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(descs || n == 0, NEF_E_NULL);
    NEF_REQUIRE(n >= 0, NEF_E_SHAPE);
    nef_pack_desc h2[PACK_MULTI_MAX];      // split-fp16 operands (wino == 3) are packed by conv_h2.hip
    int n_h2 = 0;
    for (int i = 0; i < n; ++i) {
        const nef_pack_desc& d = descs[i];
        NEF_REQUIRE(d.w && d.wp, NEF_E_NULL);
        NEF_REQUIRE(d.G > 0 && d.Cog > 0 && d.Cig > 0 && d.K > 0 && d.wino >= 0 && d.wino <= 3 &&
                        (!d.wino || d.K == 3 || d.K == 7 || (d.wino == 3 && d.K == 1)), NEF_E_SHAPE);
        NEF_REQUIRE(d.src_mode == 0 || d.wino == 3, NEF_E_UNSUPPORTED);      // synthesized sources: split-fp16 operands only
        if (d.wino == 3) {
            if (n_h2 == PACK_MULTI_MAX) {
                if (int e = nef_h2_pack(h2, n_h2, (hipStream_t)stream)) return e;
                n_h2 = 0;
            }
            h2[n_h2++] = d;
        }
    }
    if (n_h2) {
        if (int e = nef_h2_pack(h2, n_h2, (hipStream_t)stream)) return e;
    }
    for (int i0 = 0; i0 < n;) {
        PackTable t;
        int m = 0;
        int64_t biggest = 1;
        for (; i0 < n && m < PACK_MULTI_MAX; ++i0) {
            if (descs[i0].wino == 3) continue;
            t.d[m++] = descs[i0];
        }
        if (m == 0) break;
        for (int i = 0; i < m; ++i) {
            const int64_t e = (int64_t)t.d[i].G * t.d[i].Cog * t.d[i].Cig * (t.d[i].wino ? 1 : t.d[i].K);
            if (e > biggest) biggest = e;
        }
        int gx = (int)nef_cdiv(biggest, 256);
        if (gx > 64) gx = 64;
        hipLaunchKernelGGL(pack_weights_multi_kernel, dim3((unsigned)gx, (unsigned)m), dim3(256), 0, (hipStream_t)stream, t);
    }
    return nef_launch_status();
}

size_t nef_conv_args_bytes(void) { return sizeof(nef_conv_args); }

#ifdef NEF_TRACE
int nef_debug_set_trace(void* p) {
    unsigned long long* v = (unsigned long long*)p;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(nef_trace_ptr), &v, sizeof(v));
}
#endif

int nef_conv_fwd(const nef_conv_args* a, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(a && a->x && a->wp && a->y, NEF_E_NULL);
    NEF_REQUIRE(a->B > 0 && a->T > 0 && a->G > 0, NEF_E_SHAPE);
    const int K = a->K;
    NEF_REQUIRE(K == 1 || K == 3 || K == 7, NEF_E_SHAPE);
    NEF_REQUIRE(a->Cout_g % 64 == 0 && a->Cin_g > 0, NEF_E_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    NEF_REQUIRE((!a->stats && !a->bnb_slots) || a->wino == 2 || a->wino == 3, NEF_E_UNSUPPORTED);   // only the F(4,3) and split-fp16 epilogues leave slot sums
    NEF_REQUIRE(!(a->stats && a->bnb_slots), NEF_E_UNSUPPORTED);
    NEF_REQUIRE(!a->bnb_slots || (a->bnb_x && a->bnb_mean && a->bnb_invstd && a->bnb_a && a->bnb_b), NEF_E_NULL);
    NEF_REQUIRE(!a->bnb_slots || a->bnb_Bp > 0, NEF_E_SHAPE);
    NEF_REQUIRE(!a->bnb_slots || !a->bnb_up || a->T % 4 == 0, NEF_E_SHAPE);
    NEF_REQUIRE((!a->res_scale && !a->gate_rowscale && a->stats_mode == 0) || a->wino == 3, NEF_E_UNSUPPORTED);
    if (a->wino == 3) {      // operand packed by nef_pack_weight_h2: direct conv on exact fp16 splits (conv_h2.hip)
        NEF_REQUIRE(nef_h2_ok(a), NEF_E_SHAPE);
        return nef_h2_launch(a, st);
    }
    bool big = (a->Cout_g % 128 == 0);
    if (big) {
        // small problems (reference-native batch 32, L=512): a 128-row tile gives fewer workgroups than the chip has
        // CUs; halve the M tile so twice as many workgroups share the same work
        const ColTiling ct0 = make_tiling(a->B, a->T, NT);
        if ((int64_t)a->G * (a->Cout_g / 128) * ct0.n_tiles < 384) big = false;
    }
    // input channels are staged in chunks: FwdStage<K, TM>::KC (8 for the 128-row K = 7 / K = 3 kernels, 16 for their
    // 64-row variants, 64 for K = 1) -- that, not more, is what Cin_g must be a multiple of
    const int KC = K == 1 ? FwdStage<1, 1>::KC : (big ? FwdStage<3, 2>::KC : FwdStage<3, 1>::KC);
    static_assert(FwdStage<7, 2>::KC == FwdStage<3, 2>::KC && FwdStage<7, 1>::KC == FwdStage<3, 1>::KC, "stage sizes");
    NEF_REQUIRE(a->Cin_g % KC == 0, NEF_E_SHAPE);
    if (a->wino == 2) return nef_mfma_wino4_fwd(a, st);      // weights packed for F(4,3): part 2 of this file
    if (a->wino) return nef_mfma_wino_fwd(a, st);            // weights packed for F(2,3): part 3 of this file
    if (a->pro_mode != 0) {
        NEF_REQUIRE(K == 3 && a->pro_mode >= 1 && a->pro_mode <= 3 && !a->in_scale, NEF_E_UNSUPPORTED);
        NEF_REQUIRE(!(a->pro_mode & 1) || (a->pro_a && a->pro_b && a->pro_Bp > 0), NEF_E_NULL);
        NEF_REQUIRE(!(a->pro_mode & 2) || (a->T % 2 == 0), NEF_E_SHAPE);
        if (a->pro_mode & 1) {       // a column tile must not straddle two passes
            const ColTiling ct = make_tiling(a->B, a->T, NT);
            NEF_REQUIRE(ct.nseg == 1 || a->pro_Bp % ct.nseg == 0, NEF_E_SHAPE);
        }
        switch (a->pro_mode) {
            case 1: return big ? launch_conv_fwd<3, 2, 1>(*a, st) : launch_conv_fwd<3, 1, 1>(*a, st);
            case 2: return big ? launch_conv_fwd<3, 2, 2>(*a, st) : launch_conv_fwd<3, 1, 2>(*a, st);
            default: return big ? launch_conv_fwd<3, 2, 3>(*a, st) : launch_conv_fwd<3, 1, 3>(*a, st);
        }
    }
    switch (K) {
        case 7: return big ? launch_conv_fwd<7, 2>(*a, st) : launch_conv_fwd<7, 1>(*a, st);
        case 3: return big ? launch_conv_fwd<3, 2>(*a, st) : launch_conv_fwd<3, 1>(*a, st);
        default: return big ? launch_conv_fwd<1, 2>(*a, st) : launch_conv_fwd<1, 1>(*a, st);
    }
}

int nef_conv_stats_slots(int T, int Cout_g) {
    if (T <= 0 || T % 2 != 0 || Cout_g <= 0 || Cout_g % 64 != 0) return 0;
    const bool wide = (Cout_g % 128 == 0);
    if (T < (wide ? 128 : 256)) return 0;
    const int nto = wide ? 128 : 256;                    // columns per workgroup; a slot is one wave's 128 of them
    return ((T + nto - 1) / nto) * (nto / 128);
}

size_t nef_conv_bwd_weight_ws_bytes(int B, int T, int G, int Cin_g, int Cout_g, int K) {
    BwdWeightPlan p;
    if (!plan_bwd_weight(B, T, G, Cin_g, Cout_g, K, &p)) return 0;
    return (size_t)p.S * G * K * Cout_g * Cin_g * sizeof(float);
}

int nef_conv_bwd_weight(const float* x, int64_t x_bs, int64_t x_gs, const float* in_scale, int64_t sc_bs,
                        int64_t sc_gs, const float* gy, int64_t gy_bs, int64_t gy_gs, float* gw, void* ws,
                        size_t ws_bytes, int B, int T, int G, int Cin_g, int Cout_g, int K, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(x && gy && gw && ws, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && T > 0 && G > 0, NEF_E_SHAPE);
    BwdWeightPlan p;
    NEF_REQUIRE(plan_bwd_weight(B, T, G, Cin_g, Cout_g, K, &p), NEF_E_SHAPE);
    const size_t need = (size_t)p.S * G * K * Cout_g * Cin_g * sizeof(float);
    NEF_REQUIRE(ws_bytes >= need, NEF_E_WORKSPACE);
    hipStream_t st = (hipStream_t)stream;
    float* wsf = (float*)ws;
    int rc;
#define NEF_BW(KK, WCO, TCI)                                                                                          \
    rc = launch_bwd_weight<KK, WCO, TCI>(p, x, x_bs, x_gs, in_scale, sc_bs, sc_gs, gy, gy_bs, gy_gs, wsf, B, T, G,    \
                                         Cin_g, Cout_g, st)
    if (K == 7) {
        if (p.wco == 4) NEF_BW(7, 4, 1); else NEF_BW(7, 2, 1);
    } else if (K == 3) {
        if (p.wco == 4) { if (p.tci == 2) NEF_BW(3, 4, 2); else NEF_BW(3, 4, 1); }
        else { if (p.tci == 2) NEF_BW(3, 2, 2); else NEF_BW(3, 2, 1); }
    } else {
        if (p.wco == 4) { if (p.tci == 2) NEF_BW(1, 4, 2); else NEF_BW(1, 4, 1); }
        else { if (p.tci == 2) NEF_BW(1, 2, 2); else NEF_BW(1, 2, 1); }
    }
#undef NEF_BW
    if (rc != NEF_OK) return rc;
    const int64_t n = (int64_t)G * K * Cout_g * Cin_g;
    hipLaunchKernelGGL(conv_bwd_weight_reduce, dim3(nef_stream_grid(n, 256)), dim3(256), 0, st, wsf, gw, G, Cout_g,
                       Cin_g, K, p.S);
    return nef_launch_status();
}

int nef_conv_bwd_weight_pro(const float* x, int64_t x_bs, int64_t x_gs, const float* pro_a, const float* pro_b,
                            int pro_mode, int pro_Bp, const float* gy, int64_t gy_bs, int64_t gy_gs, float* gw, void* ws,
                            size_t ws_bytes, int B, int T, int G, int Cin_g, int Cout_g, int K, nef_stream_t stream) {
    NEF_ENTER();
    if (pro_mode == 0)
        return nef_conv_bwd_weight(x, x_bs, x_gs, nullptr, 0, 0, gy, gy_bs, gy_gs, gw, ws, ws_bytes, B, T, G, Cin_g, Cout_g, K,
                                   stream);
    NEF_REQUIRE(x && gy && gw && ws, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && T > 0 && G > 0, NEF_E_SHAPE);
    NEF_REQUIRE(K == 3 && pro_mode >= 1 && pro_mode <= 3, NEF_E_UNSUPPORTED);
    NEF_REQUIRE(!(pro_mode & 1) || (pro_a && pro_b && pro_Bp > 0), NEF_E_NULL);
    NEF_REQUIRE(!(pro_mode & 2) || (T % 2 == 0), NEF_E_SHAPE);
    BwdWeightPlan p;
    NEF_REQUIRE(plan_bwd_weight(B, T, G, Cin_g, Cout_g, K, &p, pro_mode), NEF_E_SHAPE);
    NEF_REQUIRE(!(pro_mode & 1) || p.ct.nseg == 1 || pro_Bp % p.ct.nseg == 0, NEF_E_SHAPE);
    const size_t need = (size_t)p.S * G * K * Cout_g * Cin_g * sizeof(float);
    NEF_REQUIRE(ws_bytes >= need, NEF_E_WORKSPACE);
    hipStream_t st = (hipStream_t)stream;
    float* wsf = (float*)ws;
    int rc = NEF_E_UNSUPPORTED;
#define NEF_BWP(WCO, TCI, PRO)                                                                                          \
    rc = launch_bwd_weight<3, WCO, TCI, PRO>(p, x, x_bs, x_gs, nullptr, 0, 0, gy, gy_bs, gy_gs, wsf, B, T, G, Cin_g,    \
                                             Cout_g, st, pro_a, pro_b, pro_Bp)
#define NEF_BWP_MODE(WCO, TCI)                                                                                          \
    {                                                                                                                 \
        if (pro_mode == 1) NEF_BWP(WCO, TCI, 1);                                                                      \
        else if (pro_mode == 2) NEF_BWP(WCO, TCI, 2);                                                                 \
        else NEF_BWP(WCO, TCI, 3);                                                                                    \
    }
    if (p.wco == 4) { if (p.tci == 2) NEF_BWP_MODE(4, 2) else NEF_BWP_MODE(4, 1) }
    else NEF_BWP_MODE(2, 1)
#undef NEF_BWP_MODE
#undef NEF_BWP
    if (rc != NEF_OK) return rc;
    const int64_t n = (int64_t)G * K * Cout_g * Cin_g;
    hipLaunchKernelGGL(conv_bwd_weight_reduce, dim3(nef_stream_grid(n, 256)), dim3(256), 0, st, wsf, gw, G, Cout_g,
                       Cin_g, K, p.S);
    return nef_launch_status();
}

// conv_h2w.hip: the weight gradient on exact fp16 splits of both operands
__attribute__((visibility("hidden"))) bool nef_h2w_ok(int B, int T, int Cig, int Cog, int K, int pro_mode);
__attribute__((visibility("hidden"))) int nef_h2w_splits(int B, int T, int G, int Cig, int Cog, int K, int pro_mode, int* partials);
__attribute__((visibility("hidden"))) int nef_h2w_launch(const float* x, int64_t x_bs, int64_t x_gs, const float* in_scale, int64_t sc_bs,
                                                         int64_t sc_gs, const float* pro_a, const float* pro_b, int pro_mode, int pro_Bp,
                                                         const float* gy, int64_t gy_bs, int64_t gy_gs, float* ws, int B, int T, int G,
                                                         int Cig, int Cog, int K, int S, float x_scale, float gy_scale,
                                                         const float* x_amax, const float* gy_amax, float* x_amax_next,
                                                         float* gy_amax_next, int* clamped, hipStream_t st);

size_t nef_conv_bwd_weight_h2_ws_bytes(int B, int T, int G, int Cin_g, int Cout_g, int K) {
    if (G <= 0 || !nef_h2w_ok(B, T, Cin_g, Cout_g, K, 0)) return 0;
    int partials = 0;      // the tile form (and with it the split count) may depend on the prologue: size for the largest
    for (int pm = 0; pm < (K == 3 ? 4 : 1); ++pm) {
        int p_ = 0;
        (void)nef_h2w_splits(B, T, G, Cin_g, Cout_g, K, pm, &p_);
        if (p_ > partials) partials = p_;
    }
    return (size_t)partials * G * K * Cout_g * Cin_g * sizeof(float);
}

int nef_conv_bwd_weight_h2(const float* x, int64_t x_bs, int64_t x_gs, const float* in_scale, int64_t sc_bs, int64_t sc_gs,
                           const float* pro_a, const float* pro_b, int pro_mode, int pro_Bp, const float* gy, int64_t gy_bs,
                           int64_t gy_gs, float* gw, void* ws, size_t ws_bytes, int B, int T, int G, int Cin_g, int Cout_g, int K,
                           float x_scale, float gy_scale, const float* x_amax, const float* gy_amax, float* x_amax_next,
                           float* gy_amax_next, int32_t* clamped, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(x && gy && gw && ws, NEF_E_NULL);
    NEF_REQUIRE(G > 0 && nef_h2w_ok(B, T, Cin_g, Cout_g, K, pro_mode), NEF_E_SHAPE);
    NEF_REQUIRE(!(pro_mode && in_scale), NEF_E_UNSUPPORTED);
    int partials = 0;
    const int S = nef_h2w_splits(B, T, G, Cin_g, Cout_g, K, pro_mode, &partials);
    NEF_REQUIRE(ws_bytes >= (size_t)partials * G * K * Cout_g * Cin_g * sizeof(float), NEF_E_WORKSPACE);
    hipStream_t st = (hipStream_t)stream;
    if (int rc = nef_h2w_launch(x, x_bs, x_gs, in_scale, sc_bs, sc_gs, pro_a, pro_b, pro_mode, pro_Bp, gy, gy_bs, gy_gs, (float*)ws, B,
                                T, G, Cin_g, Cout_g, K, S, x_scale, gy_scale, x_amax, gy_amax, x_amax_next, gy_amax_next, clamped, st))
        return rc;
    const int64_t n = (int64_t)G * K * Cout_g * Cin_g;
    hipLaunchKernelGGL(conv_bwd_weight_reduce, dim3(nef_stream_grid(n, 256)), dim3(256), 0, st, (const float*)ws, gw, G, Cout_g,
                       Cin_g, K, partials);
    return nef_launch_status();
}

int nef_conv_bwd_weight_wino4(const float* x, int64_t x_bs, int64_t x_gs, const float* in_scale, int64_t sc_bs,
                              int64_t sc_gs, const float* pro_a, const float* pro_b, int pro_mode, int pro_Bp,
                              const float* gy, int64_t gy_bs, int64_t gy_gs, float* gw, void* ws, size_t ws_bytes, int B,
                              int T, int G, int Cin_g, int Cout_g, int K, nef_stream_t stream) {
    return nef_mfma_bww_wino4(x, x_bs, x_gs, in_scale, sc_bs, sc_gs, pro_a, pro_b, pro_mode, pro_Bp, gy, gy_bs, gy_gs, gw, ws, ws_bytes, B, T, G, Cin_g, Cout_g, K, stream);
}

size_t nef_chan_sum_ws_bytes(int C) { return (size_t)CHAN_SUM_SPLIT * C * sizeof(double); }

int nef_chan_sum(const float* x, float* out, void* ws, size_t ws_bytes, int B, int C, int T, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(x && out && ws, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && C > 0 && T > 0, NEF_E_SHAPE);
    NEF_REQUIRE(ws_bytes >= nef_chan_sum_ws_bytes(C), NEF_E_WORKSPACE);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(chan_sum_partial, dim3(C * CHAN_SUM_SPLIT), dim3(256), 0, st, x, (double*)ws, B, C, T,
                       CHAN_SUM_SPLIT);
    hipLaunchKernelGGL(chan_sum_final, dim3((C + 255) / 256), dim3(256), 0, st, (const double*)ws, out, C,
                       CHAN_SUM_SPLIT);
    return nef_launch_status();
}

}  // extern "C"
#endif  // part 1

#if NEF_PART(2)
__attribute__((visibility("hidden"))) int nef_mfma_wino4_fwd(const nef_conv_args* a, hipStream_t st) {
    const int K = a->K;
        NEF_REQUIRE((K == 3 || K == 7) && a->T % 2 == 0 && a->Cin_g % WKC == 0, NEF_E_SHAPE);
        NEF_REQUIRE(a->pro_mode >= 0 && a->pro_mode <= 3 && !(a->pro_mode && a->in_scale), NEF_E_UNSUPPORTED);
        NEF_REQUIRE(K == 3 || a->pro_mode == 0, NEF_E_UNSUPPORTED);
        NEF_REQUIRE(!(a->pro_mode & 1) || (a->pro_a && a->pro_b && a->pro_Bp > 0), NEF_E_NULL);
        const bool wide = (a->Cout_g % 128 == 0);
        NEF_REQUIRE(a->T >= (wide ? 128 : 256), NEF_E_SHAPE);
        if (K == 7) return wide ? launch_conv_wino4<7, 4, 0>(*a, st) : launch_conv_wino4<7, 2, 0>(*a, st);
        switch (a->pro_mode) {
            case 0: return wide ? launch_conv_wino4<3, 4, 0>(*a, st) : launch_conv_wino4<3, 2, 0>(*a, st);
            case 1: return wide ? launch_conv_wino4<3, 4, 1>(*a, st) : launch_conv_wino4<3, 2, 1>(*a, st);
            case 2: return wide ? launch_conv_wino4<3, 4, 2>(*a, st) : launch_conv_wino4<3, 2, 2>(*a, st);
            default: return wide ? launch_conv_wino4<3, 4, 3>(*a, st) : launch_conv_wino4<3, 2, 3>(*a, st);
        }
}
#endif

#if NEF_PART(3)
__attribute__((visibility("hidden"))) int nef_mfma_wino_fwd(const nef_conv_args* a, hipStream_t st) {
    const int K = a->K;
        NEF_REQUIRE((K == 3 || K == 7) && a->T % 2 == 0 && a->Cin_g % WKC == 0, NEF_E_SHAPE);
        NEF_REQUIRE(a->pro_mode >= 0 && a->pro_mode <= 3 && !(a->pro_mode && a->in_scale), NEF_E_UNSUPPORTED);
        NEF_REQUIRE(K == 3 || a->pro_mode == 0, NEF_E_UNSUPPORTED);
        NEF_REQUIRE(!(a->pro_mode & 1) || (a->pro_a && a->pro_b && a->pro_Bp > 0), NEF_E_NULL);
        const bool wide = (a->Cout_g % 128 == 0);
        NEF_REQUIRE(a->T >= (wide ? 128 : 256), NEF_E_SHAPE);
        if (K == 7) return wide ? launch_conv_wino<7, 2, 0>(*a, st) : launch_conv_wino<7, 1, 0>(*a, st);
        switch (a->pro_mode) {
            case 0: return wide ? launch_conv_wino<3, 2, 0>(*a, st) : launch_conv_wino<3, 1, 0>(*a, st);
            case 1: return wide ? launch_conv_wino<3, 2, 1>(*a, st) : launch_conv_wino<3, 1, 1>(*a, st);
            case 2: return wide ? launch_conv_wino<3, 2, 2>(*a, st) : launch_conv_wino<3, 1, 2>(*a, st);
            default: return wide ? launch_conv_wino<3, 2, 3>(*a, st) : launch_conv_wino<3, 1, 3>(*a, st);
        }
}
#endif

#if NEF_PART(4)
extern "C" {
// conv_bww_glds.hip: the same forms with the tiles streamed by LDS-DMA through a ring of LDS buffers
__attribute__((visibility("hidden"))) bool nef_bww_glds_ok(int B, int T, int Cig, int Cog, int K, int pro_mode, int pro_Bp, bool in_scale);
__attribute__((visibility("hidden"))) int nef_bww_glds_launch(const float* x, int64_t x_bs, int64_t x_gs, const float* gy, int64_t gy_bs, int64_t gy_gs, float* ws,
                        int B, int T, int G, int Cig, int Cog, int K, int half, const float* pro_a, const float* pro_b,
                        int pro_mode, int pro_Bp, int S_max, int fixed_S, int* S_used, hipStream_t st);
}
__attribute__((visibility("hidden"))) int nef_mfma_bww_wino4(const float* x, int64_t x_bs, int64_t x_gs, const float* in_scale, int64_t sc_bs,
                              int64_t sc_gs, const float* pro_a, const float* pro_b, int pro_mode, int pro_Bp,
                              const float* gy, int64_t gy_bs, int64_t gy_gs, float* gw, void* ws, size_t ws_bytes, int B,
                              int T, int G, int Cin_g, int Cout_g, int K, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(x && gy && gw && ws, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && T >= WT && T % 2 == 0 && G > 0 && (K == 3 || K == 7), NEF_E_SHAPE);
    NEF_REQUIRE(pro_mode >= 0 && pro_mode <= 3 && (K == 3 || pro_mode == 0) && !(pro_mode && in_scale), NEF_E_UNSUPPORTED);
    NEF_REQUIRE(!(pro_mode & 1) || (pro_a && pro_b && pro_Bp > 0), NEF_E_NULL);
    BwdWeightPlan p;
    NEF_REQUIRE(plan_bwd_weight(B, T, G, Cin_g, Cout_g, K, &p, pro_mode, K == 7 ? 4 : 2), NEF_E_SHAPE);
    NEF_REQUIRE(p.ct.nseg == 1, NEF_E_SHAPE);
    const size_t need = (size_t)p.S * G * K * Cout_g * Cin_g * sizeof(float);
    NEF_REQUIRE(ws_bytes >= need, NEF_E_WORKSPACE);
    hipStream_t st = (hipStream_t)stream;
    float* wsf = (float*)ws;
    int rc = NEF_E_UNSUPPORTED;
#define NEF_BW4(WCO, PRO)                                                                                              \
    rc = launch_bwd_weight<3, WCO, 1, PRO, 2>(p, x, x_bs, x_gs, in_scale, sc_bs, sc_gs, gy, gy_bs, gy_gs, wsf, B, T, G,  \
                                              Cin_g, Cout_g, st, pro_a, pro_b, pro_Bp)
#define NEF_BW4_MODE(WCO)                                                                                              \
    {                                                                                                                 \
        if (pro_mode == 0) NEF_BW4(WCO, 0);                                                                           \
        else if (pro_mode == 1) NEF_BW4(WCO, 1);                                                                      \
        else if (pro_mode == 2) NEF_BW4(WCO, 2);                                                                      \
        else NEF_BW4(WCO, 3);                                                                                         \
    }
    if (nef_bww_glds_ok(B, T, Cin_g, Cout_g, K, pro_mode, pro_Bp, in_scale != nullptr)) {
        int S_used = 0;
        if (K == 3) {
            rc = nef_bww_glds_launch(x, x_bs, x_gs, gy, gy_bs, gy_gs, wsf, B, T, G, Cin_g, Cout_g, 3, 0, pro_a, pro_b, pro_mode,
                                     pro_Bp, p.S, 0, &S_used, st);
        } else {
            rc = nef_bww_glds_launch(x, x_bs, x_gs, gy, gy_bs, gy_gs, wsf, B, T, G, Cin_g, Cout_g, 7, 4, nullptr, nullptr, 0, 1, p.S,
                                     0, &S_used, st);
            if (rc == NEF_OK)
                rc = nef_bww_glds_launch(x, x_bs, x_gs, gy, gy_bs, gy_gs, wsf, B, T, G, Cin_g, Cout_g, 7, 5, nullptr, nullptr, 0, 1,
                                         p.S, S_used, &S_used, st);
        }
        p.S = S_used;
    } else if (K == 7) {      // taps split 4 + 3 across two launches: transposed F(4,4), then transposed F(3,4)
        if (p.wco == 4) {
            rc = launch_bwd_weight<7, 4, 1, 0, 4>(p, x, x_bs, x_gs, in_scale, sc_bs, sc_gs, gy, gy_bs, gy_gs, wsf, B, T, G, Cin_g,
                                                  Cout_g, st);
            if (rc == NEF_OK)
                rc = launch_bwd_weight<7, 4, 1, 0, 5>(p, x, x_bs, x_gs, in_scale, sc_bs, sc_gs, gy, gy_bs, gy_gs, wsf, B, T, G,
                                                      Cin_g, Cout_g, st, nullptr, nullptr, 1, p.S);
        } else {
            rc = launch_bwd_weight<7, 2, 1, 0, 4>(p, x, x_bs, x_gs, in_scale, sc_bs, sc_gs, gy, gy_bs, gy_gs, wsf, B, T, G, Cin_g,
                                                  Cout_g, st);
            if (rc == NEF_OK)
                rc = launch_bwd_weight<7, 2, 1, 0, 5>(p, x, x_bs, x_gs, in_scale, sc_bs, sc_gs, gy, gy_bs, gy_gs, wsf, B, T, G,
                                                      Cin_g, Cout_g, st, nullptr, nullptr, 1, p.S);
        }
    } else if (p.wco == 4) NEF_BW4_MODE(4) else NEF_BW4_MODE(2)
#undef NEF_BW4_MODE
#undef NEF_BW4
    if (rc != NEF_OK) return rc;
    const int64_t n = (int64_t)G * K * Cout_g * Cin_g;
    hipLaunchKernelGGL(conv_bwd_weight_reduce, dim3(nef_stream_grid(n, 256)), dim3(256), 0, st, wsf, gw, G, Cout_g,
                       Cin_g, K, p.S);
    return nef_launch_status();
}
#endif
