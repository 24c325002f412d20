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

__global__ void stem_bwd_weight_reduce(const float* __restrict__ part, float* __restrict__ gw, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int sp = 0; sp < BW_SPLIT; ++sp) s += part[(int64_t)sp * n + i];
    gw[i] = s;
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

size_t nef_stem_bwd_ws_bytes(int V) { return (size_t)BW_SPLIT * V * CPL * KW * sizeof(float); }

int nef_stem_bwd_weight(const float* x, const float* w, const float* gy, float* gw, void* ws, size_t ws_bytes, int B,
                        int V, int L, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(x && w && gy && gw && ws, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && V > 0 && L >= 4 && L % 4 == 0, NEF_E_SHAPE);
    NEF_REQUIRE(ws_bytes >= nef_stem_bwd_ws_bytes(V), NEF_E_WORKSPACE);
    const int T = L / 4;
    const int tiles = (T + BW_TP - 1) / BW_TP;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(stem_bwd_weight_kernel, dim3((unsigned)(BW_SPLIT * V * (CPL / BW_CPB))), dim3(256), 0, st, x, w,
                       gy, (float*)ws, B, V, L, T, tiles);
    const int n = V * CPL * KW;
    hipLaunchKernelGGL(stem_bwd_weight_reduce, dim3((n + 255) / 256), dim3(256), 0, st, (const float*)ws, gw, n);
    return nef_launch_status();
}

}  // extern "C"
