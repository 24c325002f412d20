// Stem of the per-lead encoder, fused: Conv1d(1->128 per lead, k15, s2, p7, no bias) -> ReLU -> MaxPool1d(3,2,1).
// Replaces reference codes/network/encoder/encoder.py:35-38 (conv1/relu/maxpool built at
// codes/network/encoder/resnet_1d.py:102-105).  HBM-bound (AI ~7 FLOP/B): the [B,128V,L/2] conv output is never
// materialised; the backward recomputes it from the 19-sample input window each lane already holds.  (In practice
// the two kernels are bound by vector-instruction issue -- see the note above stem_fwd_kernel.)
#include "nef_common.h"

namespace {

constexpr int KW = 15;        // taps
constexpr int TP = 64;        // pooled outputs per workgroup row (one per lane)
constexpr int CPL = 128;      // channels per lead

// Conv output j (length L/2) reads x[2j-7 .. 2j+7]; pooled output tp covers j in {2tp-1, 2tp, 2tp+1}.
// So lane tp needs x[4tp-9 .. 4tp+9]  (19 samples), kept in registers.
// Branch-free: one buffer-descriptor load per sample, out-of-range positions carry NEF_OOB and read as 0.0 (the hardware
// range check) -- 19 independent loads in flight.  (The first version guarded every sample with its own branch: 19 basic
// blocks per window, each load waited for before the next was issued.)
__device__ __forceinline__ void load_window(__amdgpu_buffer_rsrc_t xrow, int L, int tp, float (&xw)[19]) {
    const int base = 4 * tp - 9;
#pragma unroll
    for (int i = 0; i < 19; ++i) {
        const int p = base + i;
        xw[i] = nef_buf_f32(xrow, (p >= 0 && p < L) ? (unsigned)(p * 4) : NEF_OOB, 0);
    }
}

// Both kernels are bound by vector instructions, not by HBM (PMC: the SIMDs issue every cycle), so the conv value a
// pooled output shares with its neighbour is computed once: c(2tp-1) of lane tp IS c(2tp+1) of lane tp-1 -- same taps, same
// samples, same FMA order, hence the same bits -- and comes over the cross-lane network.  A tile therefore carries one
// halo lane on the left (lane 0 computes, does not store): FWD_TP useful outputs per 64 lanes, 30 instead of 45 FMAs each.
constexpr int FWD_TP = TP - 1;

__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       float* __restrict__ y, int B, int V, int L, int T,
                                                       int tiles_per_row) {
    __shared__ float wl[CPL * 16];   // [co][16] (15 taps + pad) for 16-byte broadcast reads
    int bid = blockIdx.x;
    const int tile = bid % tiles_per_row;
    bid /= tiles_per_row;
    const int v = bid % V;
    const int b = bid / V;
    for (int i = threadIdx.x; i < CPL * 16; i += 256) {
        const int co = i >> 4, k = i & 15;
        wl[i] = k < KW ? w[(v * CPL + co) * KW + k] : 0.f;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tp = tile * FWD_TP + lane - 1;
    float xw[19];
    load_window(nef_rsrc(x + ((int64_t)b * V + v) * L), L, tp, xw);
    __syncthreads();
    const int Lc = L / 2;
    const bool has_l = (2 * tp - 1) >= 0;        // pool padding: window positions outside [0, Lc) are -inf
    const bool has_r = (2 * tp + 1) < Lc;
    const bool owns = lane > 0 && tp < T;
    float* yrow = y + ((int64_t)b * V * CPL + (int64_t)v * CPL) * T + tp;
    for (int co = wave; co < CPL; co += 4) {
        const float4* w4 = reinterpret_cast<const float4*>(wl + co * 16);
        float wk[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 t4 = w4[q];
            wk[4 * q] = t4.x; wk[4 * q + 1] = t4.y; wk[4 * q + 2] = t4.z; wk[4 * q + 3] = t4.w;
        }
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            c1 = fmaf(wk[k], xw[k + 2], c1);    // j = 2tp   : x[4tp-7+k]
            c2 = fmaf(wk[k], xw[k + 4], c2);    // j = 2tp+1 : x[4tp-5+k]
        }
        const float c0 = __shfl_up(c2, 1);      // j = 2tp-1 : the left neighbour's j = 2(tp-1)+1
        float m = fmaxf(c1, 0.f);
        if (has_l) m = fmaxf(m, fmaxf(c0, 0.f));
        if (has_r) m = fmaxf(m, fmaxf(c2, 0.f));
        if (owns) yrow[(int64_t)co * T] = m;
    }
}

// Backward wrt the conv weight.  A workgroup owns 16 channels of one lead (4 per wave) and a strided share of the
// (sample, time-tile) space; each lane keeps 4x15 partial sums, reduced across the wave once at the end.
// The gradient is routed to the conv output the pool selected and accumulated PER CONV OUTPUT j, each owned by one lane
// (j = 2tp and j = 2tp+1; what the right neighbour routes to its j = 2tp'-1 arrives over the cross-lane network), so a
// pooled output costs 30 FMAs for the recomputed conv values and 30 for the accumulation instead of 45 + 15 + 30
// selects.  Lane 0 is a left halo (supplies c(2tp+1)), lane 63 a right halo (supplies its routed gradient): BW_TP useful
// outputs per 64 lanes.
constexpr int BW_CPW = 4;                  // channels per wave
constexpr int BW_CPB = 4 * BW_CPW;         // channels per workgroup
constexpr int BW_SPLIT = 64;
constexpr int BW_TP = TP - 2;

__global__ __launch_bounds__(256) void stem_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ gy, float* __restrict__ part,
                                                              int B, int V, int L, int T, int tiles_per_row) {
    int bid = blockIdx.x;
    const int cg = bid % (CPL / BW_CPB);
    bid /= (CPL / BW_CPB);
    const int v = bid % V;
    const int split = bid / V;
    const int lane = threadIdx.x & 63;
    // wave index made provably uniform: the 4x15 filter taps then sit in scalar registers, not in 60 VGPRs
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ch0 = v * CPL + cg * BW_CPB + wave * BW_CPW;
    float wk[BW_CPW][KW];
#pragma unroll
    for (int c = 0; c < BW_CPW; ++c)
#pragma unroll
        for (int k = 0; k < KW; ++k) wk[c][k] = w[(ch0 + c) * KW + k];
    float acc[BW_CPW][KW];
#pragma unroll
    for (int c = 0; c < BW_CPW; ++c)
#pragma unroll
        for (int k = 0; k < KW; ++k) acc[c][k] = 0.f;
    const int Lc = L / 2;
    const int n_units = B * tiles_per_row;
    // Register double-buffering across units: the window and the four gradients of unit u+1 are requested before the 240
    // FMAs of unit u, so their L2 / HBM latency hides under arithmetic instead of in front of it.
    float xn[19], gn[BW_CPW];
#define NEF_STEM_FETCH(UNIT)                                                                                           \
    {                                                                                                                 \
        const int b_ = (UNIT) / tiles_per_row;                                                                        \
        const int tp_ = ((UNIT) - b_ * tiles_per_row) * BW_TP + lane - 1;                                             \
        load_window(nef_rsrc(x + ((int64_t)b_ * V + v) * L), L, tp_, xn);                                             \
        const __amdgpu_buffer_rsrc_t grs_ = nef_rsrc(gy + ((int64_t)b_ * V * CPL + ch0) * T);                         \
        const unsigned go_ = (tp_ >= 0 && tp_ < T) ? (unsigned)(tp_ * 4) : NEF_OOB;                                   \
        _Pragma("unroll") for (int c = 0; c < BW_CPW; ++c) gn[c] = nef_buf_f32(grs_, go_, (unsigned)(c * T * 4));     \
    }
#ifndef NEF_STEM_PREFETCH
#define NEF_STEM_PREFETCH 1
#endif
    if (NEF_STEM_PREFETCH && split < n_units) NEF_STEM_FETCH(split)
    for (int unit = split; unit < n_units; unit += BW_SPLIT) {
        if (!NEF_STEM_PREFETCH) NEF_STEM_FETCH(unit)
        const int b = unit / tiles_per_row;
        const int tp = (unit - b * tiles_per_row) * BW_TP + lane - 1;
        float xw[19], gc[BW_CPW];
#pragma unroll
        for (int i = 0; i < 19; ++i) xw[i] = xn[i];
#pragma unroll
        for (int c = 0; c < BW_CPW; ++c) gc[c] = gn[c];
        if (NEF_STEM_PREFETCH && unit + BW_SPLIT < n_units) NEF_STEM_FETCH(unit + BW_SPLIT)
        const bool valid = tp >= 0 && tp < T;
        const bool owns = valid && lane > 0 && lane < 63;
        const bool has_l = (2 * tp - 1) >= 0;
        const bool has_r = (2 * tp + 1) < Lc;
#pragma unroll
        for (int c = 0; c < BW_CPW; ++c) {
            const float g = gc[c];
            float c1 = 0.f, c2 = 0.f;
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                c1 = fmaf(wk[c][k], xw[k + 2], c1);
                c2 = fmaf(wk[c][k], xw[k + 4], c2);
            }
            const float c0 = __shfl_up(c2, 1);
            // arg-max over the pool window in scan order (first maximum wins), then the ReLU gate
            float best = has_l ? fmaxf(c0, 0.f) : -INFINITY;
            float pre = c0;
            int sel = 0;
            {
                const float r1 = fmaxf(c1, 0.f);
                if (r1 > best) { best = r1; sel = 1; pre = c1; }
            }
            if (has_r) {
                const float r2 = fmaxf(c2, 0.f);
                if (r2 > best) { best = r2; sel = 2; pre = c2; }
            }
            const float ge = pre > 0.f ? g : 0.f;
            const float from_right = __shfl_down(sel == 0 ? ge : 0.f, 1);      // the neighbour's j = 2tp'-1 is my 2tp+1
            const float g1 = owns && sel == 1 ? ge : 0.f;
            const float g2 = owns ? (sel == 2 ? ge : 0.f) + from_right : 0.f;
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                acc[c][k] = fmaf(g1, xw[k + 2], acc[c][k]);
                acc[c][k] = fmaf(g2, xw[k + 4], acc[c][k]);
            }
        }
    }
#undef NEF_STEM_FETCH
#pragma unroll
    for (int c = 0; c < BW_CPW; ++c)
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            const float s = nef_wave_sum(acc[c][k]);
            if (lane == 0) part[((int64_t)split * V * CPL + ch0 + c) * KW + k] = s;
        }
}

// ------------------------------------------------------------------------------------------------------------
// Round 3: the same gradient on the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, an fmaf chain over k).
// The VALU kernel above spends 60 FMAs + ~40 routing instructions per (pooled output, channel) and is bound by vector
// issue (~490 us at config 2: 400 vector instructions per 62-output unit and wave, 6 waves per SIMD) -- the kernel is
// COMPUTE-bound, not HBM-bound: 14.8 GFLOP over 0.51 GB is 29 FLOP/B, above the 19.7 FLOP/B ridge of the fp32 pipes.
// The matrix pipe runs beside the vector pipe, so both GEMMs of the gradient move there and the vector units keep only
// the pool / ReLU routing:
//   GEMM 1 (recompute, transposed so that no data has to move between the two GEMMs):
//       C_i^T[tp][co] = sum_tap x[4tp - 9 + 2i + tap] * w[co][tap],   i = 0, 1, 2  (conv outputs j = 2tp - 1 + i)
//       A = the x window (m = tp, k = tap), B = w^T (k = tap, n = co); 4 k-steps of 4 taps (tap 15 is a zero weight).
//       A lane then holds C_i for ONE channel (n = lane % 16) and the four pooled outputs tp = 4 * (lane / 16) + r.
//   routing (vector): arg-max over (c0, c1, c2) in scan order, ReLU gate, g_i[tp][co] = gy[tp][co] if i was selected.
//   GEMM 2: gW^T[tap][co] += sum_tp x[4tp - 9 + 2i + tap] * g_i[tp][co]: A = x (m = tap, k = tp), B = g_i -- the reduction
//       index of an MFMA may be enumerated in any order as long as A and B agree, and k-step r with tp = 4 * (lane / 16)
//       + r is exactly register r of the routing result: B comes straight out of the accumulator layout of GEMM 1.
// A workgroup owns one (sample, lead) row: the 5000 input samples sit in LDS (zero-filled halo), every A fragment is a
// conflict-free ds_read_b32 shared by the wave's two 16-channel tiles; 24 matrix instructions per (16 outputs x 16
// channels).  One partial [128][15] per row slot, reduced by stem_bwd_weight_reduce.
constexpr int MB_SPLIT = 256;              // partial slots: rows b = slot, slot + 256, ...
constexpr int MB_HALO = 16;                // LDS index of x[0]
constexpr int MB_TAIL = 96;                // the last 16-output tile reads up to x[L + 66]
typedef float f32x4s __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stem_bwd_weight_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                   const float* __restrict__ gy, float* __restrict__ part,
                                                                   int B, int V, int L, int T) {
    extern __shared__ float xs[];          // [MB_HALO + L + MB_TAIL]: xs[MB_HALO + p] = x[p], zeros outside [0, L)
    const int v = blockIdx.x % V, slot = blockIdx.x / V;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n16 = lane & 15, kk = lane >> 4;
    const int Lc = L / 2;
    const int xlen = MB_HALO + L + MB_TAIL;
    // B operand of GEMM 1: w^T[tap = 4s + kk][co], two 16-channel tiles per wave (channels 32 * wave + 16 * ct + n16)
    float wb[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int tap = 4 * s + kk;
            wb[ct][s] = tap < KW ? w[(v * CPL + 32 * wave + 16 * ct + n16) * KW + tap] : 0.f;
        }
    f32x4s acc[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) acc[ct] = f32x4s{0.f, 0.f, 0.f, 0.f};
    for (int b = slot; b < B; b += MB_SPLIT) {
        __syncthreads();                   // the previous row's readers are done
        const float* xrow = x + ((int64_t)b * V + v) * L;
        for (int i = threadIdx.x; i < xlen; i += 256) {
            const int p = i - MB_HALO;
            xs[i] = (p >= 0 && p < L) ? xrow[p] : 0.f;
        }
        __syncthreads();
        const __amdgpu_buffer_rsrc_t grs = nef_rsrc(gy + ((int64_t)b * V * CPL + v * CPL + 32 * wave) * T);     // wave-uniform
        // gradients of the lane's four pooled outputs of its two channels (16 bytes each; rows are 8-byte aligned), fetched one
        // tile ahead; the ragged last tile of a row goes element by element
#define NEF_STEM_GLOAD(TP0, DST)                                                                                       \
    _Pragma("unroll") for (int ct_ = 0; ct_ < 2; ++ct_) {                                                           \
        const int tpl_ = (TP0) + 4 * kk;                                                                            \
        DST[ct_] = nef_buf_f32x4(grs, tpl_ + 3 < T ? (unsigned)((n16 * T + tpl_) * 4) : NEF_OOB, (unsigned)(16 * ct_ * T * 4)); \
        if (tpl_ + 3 >= T && tpl_ < T) {                                                                            \
            _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_)                                                        \
                DST[ct_][r_] = nef_buf_f32(grs, tpl_ + r_ < T ? (unsigned)((n16 * T + tpl_ + r_) * 4) : NEF_OOB,    \
                                           (unsigned)(16 * ct_ * T * 4));                                           \
        }                                                                                                           \
    }
        f32x4s gnext[2];
        NEF_STEM_GLOAD(0, gnext)
        for (int tp0 = 0; tp0 < T; tp0 += 16) {
            f32x4s gcur[2] = {gnext[0], gnext[1]};
            if (tp0 + 16 < T) NEF_STEM_GLOAD(tp0 + 16, gnext)
            // A fragments: GEMM 1 lane (m = tp = n16, k = kk) reads x[4(tp0 + n16) - 9 + o + kk], o = 2i + 4s in {0, 2, .., 16};
            // GEMM 2 lane (m = tap = n16, k = kk) reads x[4(tp0 + 4kk + r) - 9 + o' + n16], o' = 2i in {0, 2, 4}
            const float* p1 = xs + MB_HALO + 4 * (tp0 + n16) - 9 + kk;
            const float* p2 = xs + MB_HALO + 4 * (tp0 + 4 * kk) - 9 + n16;
            float a1[9], a2[9];
#pragma unroll
            for (int o = 0; o < 9; ++o) {
                a1[o] = p1[2 * o];
                a2[o] = p2[2 * o];
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int tpl = tp0 + 4 * kk;
                const f32x4s g4 = gcur[ct];
                f32x4s c[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    c[i] = f32x4s{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i + 2 * s], wb[ct][s], c[i], 0, 0, 0);
                }
                f32x4s gsel[3];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int tp = tpl + r;
                    const bool has_l = (2 * tp - 1) >= 0, has_r = (2 * tp + 1) < Lc;
                    // arg-max over the pool window in scan order (first maximum wins), then the ReLU gate -- as the forward
                    float best = has_l ? fmaxf(c[0][r], 0.f) : -INFINITY;
                    float pre = c[0][r];
                    int sel = 0;
                    const float r1 = fmaxf(c[1][r], 0.f);
                    if (r1 > best) { best = r1; sel = 1; pre = c[1][r]; }
                    if (has_r) {
                        const float r2 = fmaxf(c[2][r], 0.f);
                        if (r2 > best) { best = r2; sel = 2; pre = c[2][r]; }
                    }
                    const float ge = (tp < T && pre > 0.f) ? g4[r] : 0.f;
                    gsel[0][r] = sel == 0 ? ge : 0.f;
                    gsel[1][r] = sel == 1 ? ge : 0.f;
                    gsel[2][r] = sel == 2 ? ge : 0.f;
                }
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[i + 2 * r], gsel[i][r], acc[ct], 0, 0, 0);
            }
        }
    }
#undef NEF_STEM_GLOAD
    // acc[ct][r] = gW^T[tap = 4kk + r][co = n16]
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tap = 4 * kk + r;
            if (tap < KW) part[((int64_t)slot * V * CPL + v * CPL + 32 * wave + 16 * ct + n16) * KW + tap] = acc[ct][r];
        }
}

// gw[i] = sum over the slots of part[slot][i], fixed order (deterministic).  Round 6: 64 outputs per workgroup, its four waves take
// every fourth slot each and the four sub-sums are added in wave order (round 5: one thread per output walked all 256 slots alone --
// 23 workgroups, 62 us for 5.9 MB).
__global__ __launch_bounds__(256) void stem_bwd_weight_reduce(const float* __restrict__ part, float* __restrict__ gw, int n, int nsplit) {
    __shared__ float sm[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (i < n) {      // eight loads in flight, added in slot order
        for (int sp0 = wave; sp0 < nsplit; sp0 += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = sp0 + 4 * u < nsplit ? part[(int64_t)(sp0 + 4 * u) * n + i] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
    }
    sm[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && i < n) gw[i] = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
}

}  // namespace

extern "C" {

int nef_stem_fwd(const float* x, const float* w, float* y, int B, int V, int L, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(x && w && y, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && V > 0 && L >= 4 && L % 4 == 0, NEF_E_SHAPE);
    const int T = L / 4;
    const int tiles = (T + FWD_TP - 1) / FWD_TP;
    hipLaunchKernelGGL(stem_fwd_kernel, dim3((unsigned)((int64_t)B * V * tiles)), dim3(256), 0, (hipStream_t)stream, x,
                       w, y, B, V, L, T, tiles);
    return nef_launch_status();
}

size_t nef_stem_bwd_ws_bytes(int V) { return (size_t)(MB_SPLIT > BW_SPLIT ? MB_SPLIT : BW_SPLIT) * V * CPL * KW * sizeof(float); }

int nef_stem_bwd_weight(const float* x, const float* w, const float* gy, float* gw, void* ws, size_t ws_bytes, int B,
                        int V, int L, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(x && w && gy && gw && ws, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && V > 0 && L >= 4 && L % 4 == 0, NEF_E_SHAPE);
    NEF_REQUIRE(ws_bytes >= nef_stem_bwd_ws_bytes(V), NEF_E_WORKSPACE);
    const int T = L / 4;
    hipStream_t st = (hipStream_t)stream;
    const int n = V * CPL * KW;
#ifndef NEF_STEM_MFMA
#define NEF_STEM_MFMA 1
#endif
    const size_t lds = (size_t)(MB_HALO + L + MB_TAIL) * sizeof(float);
    if (NEF_STEM_MFMA && T % 2 == 0 && lds <= 64 * 1024) {       // a whole input row in LDS; gy rows 8-byte aligned
        const int slots = B < MB_SPLIT ? B : MB_SPLIT;
        hipLaunchKernelGGL(stem_bwd_weight_mfma_kernel, dim3((unsigned)(slots * V)), dim3(256), lds, st, x, w, gy, (float*)ws, B,
                           V, L, T);
        hipLaunchKernelGGL(stem_bwd_weight_reduce, dim3((n + 63) / 64), dim3(256), 0, st, (const float*)ws, gw, n, slots);
        return nef_launch_status();
    }
    const int tiles = (T + BW_TP - 1) / BW_TP;
    hipLaunchKernelGGL(stem_bwd_weight_kernel, dim3((unsigned)(BW_SPLIT * V * (CPL / BW_CPB))), dim3(256), 0, st, x, w,
                       gy, (float*)ws, B, V, L, T, tiles);
    hipLaunchKernelGGL(stem_bwd_weight_reduce, dim3((n + 63) / 64), dim3(256), 0, st, (const float*)ws, gw, n, BW_SPLIT);
    return nef_launch_status();
}

}  // extern "C"
