// HBM-bound passes of the Nef-Net step: channel scaling, gates, lead mean / Standin mix, linear x2 upsampling,
// BatchNorm (training statistics, affine+ReLU, backward), the 64->1 output conv with sigmoid(x/3), the
// L1/L2 + Standin losses and the momentum-SGD update.  Reference call sites are cited per kernel; all are
// `codes/network/model_nefnet.py` unless stated.  One lane = consecutive time samples, so every global access is
// a coalesced 256-byte wave transaction along the time axis; reductions use wavefront shuffles.
#include "nef_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// y[b][c][t] = x[b][c][t] * s[b][c]            (:122-123, :166/:170/:174/:186)
// ------------------------------------------------------------------------------------------------
__global__ void chscale_fwd_kernel(const float* __restrict__ x, const float* __restrict__ s, int64_t s_bs,
                                   float* __restrict__ y, int B, int C, int T) {
    const int64_t rows = (int64_t)B * C;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const int b = (int)(row / C), c = (int)(row % C);
        const float f = s[(int64_t)b * s_bs + c];
        const float* xr = x + row * T;
        float* yr = y + row * T;
        for (int t = lane; t < T; t += 64) yr[t] = xr[t] * f;
    }
}

__global__ void chscale_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                   const float* __restrict__ s, int64_t s_bs, float* __restrict__ gx,
                                   float* __restrict__ gs, int B, int C, int T, int relu_x) {
    const int64_t rows = (int64_t)B * C;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const int b = (int)(row / C), c = (int)(row % C);
        const float f = s[(int64_t)b * s_bs + c];
        const float* gr = gy + row * T;
        const float* xr = x + row * T;
        float* gxr = gx + row * T;
        float acc = 0.f;
        for (int t = lane; t < T; t += 64) {
            const float g = gr[t], xv = xr[t];
            acc = fmaf(g, xv, acc);
            // relu_x: x is a ReLU output whose producer would mask this gradient first thing in its own backward
            gxr[t] = (relu_x && !(xv > 0.f)) ? 0.f : g * f;
        }
        acc = nef_wave_sum(acc);
        if (lane == 0) gs[row] = acc;
    }
}

__global__ void gate_kernel(const float* __restrict__ g, const float* __restrict__ ref, float* __restrict__ out,
                            float scale, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = ref[i] > 0.f ? g[i] * scale : 0.f;
}

// 16-byte version (n % 4 == 0, pointers 16-byte aligned)
__global__ void gate4_kernel(const nef_f32x4* __restrict__ g, const nef_f32x4* __restrict__ ref,
                             nef_f32x4* __restrict__ out, float scale, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const nef_f32x4 gv = g[i], rv = ref[i];
        nef_f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rv[e] > 0.f ? gv[e] * scale : 0.f;
        out[i] = o;
    }
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                           int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = a[i] + b[i];
}

// Time-window crop of a grouped view and its zero-filling inverse.  Used to run `z2_conv1` only on the six time samples
// around the two rows that roi_algin reads: everything else of that block is never consumed.
__global__ void window_crop_kernel(const float* __restrict__ src, int64_t x_bs, int64_t x_gs, float* __restrict__ dst,
                                   int B, int G, int Cg, int T, int t0, int W) {
    const int64_t n = (int64_t)B * G * Cg * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        int64_t r = i / W;
        const int c = (int)(r % Cg);
        r /= Cg;
        const int g = (int)(r % G);
        const int b = (int)(r / G);
        dst[i] = src[(int64_t)b * x_bs + (int64_t)g * x_gs + (int64_t)c * T + t0 + w];
    }
}

__global__ void window_scatter_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t y_bs, int64_t y_gs,
                                      int B, int G, int Cg, int T, int t0, int W) {
    const int64_t rows = (int64_t)B * G * Cg;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const int c = (int)(row % Cg);
        const int g = (int)((row / Cg) % G);
        const int b = (int)(row / ((int64_t)Cg * G));
        float* d = dst + (int64_t)b * y_bs + (int64_t)g * y_gs + (int64_t)c * T;
        const float* s = src + row * W - t0;
        for (int t = lane; t < T; t += 64) d[t] = (t >= t0 && t < t0 + W) ? s[t] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// latent = cat(mean_v z1, mean_v z2r)          (:146-151)
// ------------------------------------------------------------------------------------------------
__global__ void lead_mean_kernel(const float* __restrict__ z1, const float* __restrict__ z2r,
                                 float* __restrict__ latent, int B, int V, int T) {
    const int64_t rows = (int64_t)B * 256;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float fv = (float)V;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const int b = (int)(row >> 8), c = (int)(row & 255);
        const float* src = (c < 128 ? z1 : z2r) + ((int64_t)b * V * 128 + (c & 127)) * T;
        float* dst = latent + row * T;
        for (int t = lane; t < T; t += 64) {
            float s = src[t];
            for (int v = 1; v < V; ++v) s += src[(int64_t)v * 128 * T + t];
            dst[t] = s / fv;
        }
    }
}

// D[0] = q*latent ; D[1] = q*cat(z1[c1], latent[128:]) ; D[2] = q*cat(latent[:128], z2r[c2])     (:159-176)
// SHARED: only the two DISTINCT inputs are written, D [2B][256][T] = (q*cat(z1m, z2m) | q*cat(z1[c1], z2r[c2])): the three
// decoder inputs of model_nefnet.py:159-176 are q*cat(mean|mean), q*cat(pick|mean), q*cat(mean|pick), and the first
// decoder conv is linear in the two channel halves, so it only has to see each half once (engine.decoder_fwd)
template <bool SHARED>
__global__ void mix_fwd_kernel(const float* __restrict__ latent, const float* __restrict__ z1,
                               const float* __restrict__ z2r, const float* __restrict__ q, float* __restrict__ D,
                               int B, int V, int T, int c1, int c2, const int32_t* __restrict__ choice_dev) {
    if (choice_dev) { c1 = choice_dev[0]; c2 = choice_dev[1]; }      // graph replay: the Standin draw lives on the device
    const int64_t rows = (int64_t)B * 256;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t pass = rows * T;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const int b = (int)(row >> 8), c = (int)(row & 255);
        const float f = q[row];
        const float* lat = latent + row * T;
        const float* pick = c < 128 ? z1 + ((int64_t)b * V * 128 + c1 * 128 + c) * T
                                    : z2r + ((int64_t)b * V * 128 + c2 * 128 + (c - 128)) * T;
        float* d0 = D + row * T;
        for (int t = lane; t < T; t += 64) {
            const float l = lat[t], p = pick[t];
            d0[t] = f * l;
            if (SHARED) {
                d0[pass + t] = f * p;
            } else {
                d0[pass + t] = f * (c < 128 ? p : l);
                d0[2 * pass + t] = f * (c < 128 ? l : p);
            }
        }
    }
}

// lead_mean + mix_fwd<SHARED> in one pass: a (sample, channel) row of all V leads is read once and leaves the lead mean
// (latent), q*mean (D2 pass 0) and q*lead[c1 or c2] (D2 pass 1) -- the picked lead is one of the rows just averaged.
// Same expressions in the same order as the two kernels above: bit-identical outputs.  W floats per access (1 or 2).
template <int W>
__global__ __launch_bounds__(256) void lead_mean_mix_shared_kernel(const float* __restrict__ z1,
                                                                   const float* __restrict__ z2r,
                                                                   const float* __restrict__ q,
                                                                   float* __restrict__ latent, float* __restrict__ D2,
                                                                   int B, int V, int T, int c1, int c2,
                                                                   const int32_t* __restrict__ choice_dev) {
    typedef float vec __attribute__((ext_vector_type(W)));
    if (choice_dev) { c1 = choice_dev[0]; c2 = choice_dev[1]; }
    const int64_t rows = (int64_t)B * 256;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float fv = (float)V;
    const int TW = T / W;
    const int64_t pass = rows * TW;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const int b = (int)(row >> 8), c = (int)(row & 255);
        const vec* src = (const vec*)((c < 128 ? z1 : z2r) + ((int64_t)b * V * 128 + (c & 127)) * T);
        const int cp = c < 128 ? c1 : c2;
        const int64_t lead = (int64_t)128 * TW;
        const float f = q[row];
        vec* lat = (vec*)latent + row * TW;
        vec* d0 = (vec*)D2 + row * TW;
        // two column groups per trip: 2 V loads in flight per lane before the first add (the row loop is latency-bound
        // otherwise: one dependent load chain per trip); same expressions per element
        int t = lane;
        for (; t + 64 < TW; t += 128) {
            vec s0 = src[t], s1 = src[t + 64];
            vec p0 = s0, p1 = s1;
            for (int v = 1; v < V; ++v) {
                const vec x0 = src[(int64_t)v * lead + t], x1 = src[(int64_t)v * lead + t + 64];
                s0 += x0;
                s1 += x1;
                if (v == cp) { p0 = x0; p1 = x1; }
            }
            const vec m0 = s0 / fv, m1 = s1 / fv;
            lat[t] = m0;
            lat[t + 64] = m1;
            d0[t] = f * m0;
            d0[t + 64] = f * m1;
            d0[pass + t] = f * p0;
            d0[pass + t + 64] = f * p1;
        }
        for (; t < TW; t += 64) {
            vec s = src[t];
            vec pk = s;
            for (int v = 1; v < V; ++v) {
                const vec x = src[(int64_t)v * lead + t];
                s += x;
                if (v == cp) pk = x;
            }
            const vec m = s / fv;
            lat[t] = m;
            d0[t] = f * m;
            d0[pass + t] = f * pk;
        }
    }
}

// The same pass for T % 4 == 2 (configs[1]: T = 1250) on 16-byte accesses: a row is 5000 bytes, so every second row starts 8 bytes off a
// 16-byte boundary and the kernel above is held to 8-byte accesses (0.54-0.70 of the 16-byte rate).  Two consecutive channel rows
// (c even, c + 1) are one aligned span of T / 2 four-float vectors in every tensor involved; a wave takes such a PAIR, and the one
// vector that straddles the two rows carries both rows' scale.  Same expressions per element: bit-identical outputs.
__global__ __launch_bounds__(256) void lead_mean_mix_shared_pair_kernel(const float* __restrict__ z1, const float* __restrict__ z2r,
                                                                        const float* __restrict__ q, float* __restrict__ latent,
                                                                        float* __restrict__ D2, int B, int V, int T, int c1, int c2,
                                                                        const int32_t* __restrict__ choice_dev) {
    typedef float vec __attribute__((ext_vector_type(4)));
    if (choice_dev) { c1 = choice_dev[0]; c2 = choice_dev[1]; }
    const int64_t pairs = (int64_t)B * 128;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float fv = (float)V;
    const int TW = T / 2;                     // vectors per row pair
    const int mid = T / 4;                    // the vector with two elements of either row
    const int64_t pass = pairs * TW;
    const int64_t lead = (int64_t)128 * T / 4;
    for (int64_t pr = (int64_t)blockIdx.x * 4 + wave; pr < pairs; pr += (int64_t)gridDim.x * 4) {
        const int64_t row = 2 * pr;
        const int b = (int)(row >> 8), c = (int)(row & 255);
        const vec* src = (const vec*)((c < 128 ? z1 : z2r) + ((int64_t)b * V * 128 + (c & 127)) * T);
        const int cp = c < 128 ? c1 : c2;
        const float f0 = q[row], f1 = q[row + 1];
        vec* lat = (vec*)latent + pr * TW;
        vec* d0 = (vec*)D2 + pr * TW;
        int t = lane;
        for (; t + 64 < TW; t += 128) {
            vec s0 = src[t], s1 = src[t + 64];
            vec p0 = s0, p1 = s1;
            for (int v = 1; v < V; ++v) {
                const vec x0 = src[(int64_t)v * lead + t], x1 = src[(int64_t)v * lead + t + 64];
                s0 += x0;
                s1 += x1;
                if (v == cp) { p0 = x0; p1 = x1; }
            }
            const vec m0 = s0 / fv, m1 = s1 / fv;
            const vec g0 = t < mid ? vec{f0, f0, f0, f0} : (t > mid ? vec{f1, f1, f1, f1} : vec{f0, f0, f1, f1});
            const vec g1 = t + 64 < mid ? vec{f0, f0, f0, f0} : (t + 64 > mid ? vec{f1, f1, f1, f1} : vec{f0, f0, f1, f1});
            lat[t] = m0;
            lat[t + 64] = m1;
            d0[t] = g0 * m0;
            d0[t + 64] = g1 * m1;
            d0[pass + t] = g0 * p0;
            d0[pass + t + 64] = g1 * p1;
        }
        for (; t < TW; t += 64) {
            vec sm = src[t];
            vec pk = sm;
            for (int v = 1; v < V; ++v) {
                const vec x = src[(int64_t)v * lead + t];
                sm += x;
                if (v == cp) pk = x;
            }
            const vec m = sm / fv;
            const vec g0 = t < mid ? vec{f0, f0, f0, f0} : (t > mid ? vec{f1, f1, f1, f1} : vec{f0, f0, f1, f1});
            lat[t] = m;
            d0[t] = g0 * m;
            d0[pass + t] = g0 * pk;
        }
    }
}

__device__ __forceinline__ float up2_bwd_edge(const float* __restrict__ gr, int Tin, int m) {
    const int To = 2 * Tin;
    float s = 0.f;
#pragma unroll
    for (int d = -1; d <= 2; ++d) {
        const int i = 2 * m + d;
        if (i < 0 || i >= To) continue;
        float src = 0.5f * ((float)i + 0.5f) - 0.5f;
        if (src < 0.f) src = 0.f;
        int i0 = (int)src;
        if (i0 > Tin - 1) i0 = Tin - 1;
        const int i1 = i0 + (i0 < Tin - 1 ? 1 : 0);
        const float l1 = src - (float)i0;
        const float l0 = 1.f - l1;
        const float g = gr[i];
        if (i0 == m) s += l0 * g;
        if (i1 == m) s += l1 * g;
    }
    return s;
}

// adjoint of Upsample(x2, linear, align_corners=False) at input position m, read from the 2*Tin-long gradient row `gr`
// (the expression of upsample2_bwd_rows, so a consumer that rebuilds it on the fly is bit-identical to the two-pass form)
__device__ __forceinline__ float up2_adjoint(const float* __restrict__ gr, int Tin, int m) {
    if (m == 0 || m == Tin - 1) return up2_bwd_edge(gr, Tin, m);
    return 0.25f * gr[2 * m - 1] + 0.75f * gr[2 * m] + 0.75f * gr[2 * m + 1] + 0.25f * gr[2 * m + 2];
}

// UP: gD is the gradient wrt the x2-UPSAMPLED decoder input [3B][256][2T]; its adjoint is taken while reading
// SHARED: gD has two passes (wrt q*cat(mean|mean) and q*cat(pick|pick), see mix_fwd_kernel<true>)
template <bool UP, bool SHARED>
__global__ void mix_bwd_kernel(const float* __restrict__ gD, const float* __restrict__ latent,
                               const float* __restrict__ z1, const float* __restrict__ z2r,
                               const float* __restrict__ q, float* __restrict__ gz1, float* __restrict__ gz2r,
                               float* __restrict__ gq, int B, int V, int T, int c1, int c2,
                               const int32_t* __restrict__ choice_dev, int relu_z1) {
    if (choice_dev) { c1 = choice_dev[0]; c2 = choice_dev[1]; }
    const int64_t rows = (int64_t)B * 256;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t pass = rows * T;
    const float fv = (float)V;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const int b = (int)(row >> 8), c = (int)(row & 255);
        const bool first = c < 128;
        const int cc = c & 127;
        const int pick_v = first ? c1 : c2;
        const float f = q[row];
        const float* lat = latent + row * T;
        const float* zsrc = (first ? z1 : z2r) + ((int64_t)b * V * 128 + cc) * T;
        float* gdst = (first ? gz1 : gz2r) + ((int64_t)b * V * 128 + cc) * T;
        const float* g0 = gD + row * (UP ? 2 * T : T);
        const int64_t gpass = UP ? 2 * pass : pass;
        float acc = 0.f;
        if (SHARED && (T & 1) == 0 && T >= 8) {
            // Vector path of the train step's launch (two passes, x2 adjoint, T even): a lane owns two consecutive
            // positions (t, t+1), t even: the 2T-long gradient rows are read as one 16-byte word g[2t..2t+3] plus the two
            // neighbours, everything else as 8-byte words.  Same expressions as the scalar path below (gz1 / gz2r are
            // bit-identical; gq sums the same terms in a different lane order); the first and the last pair contain an edge
            // position and take the scalar formulas.
            const float* g1r = g0 + gpass;
            for (int p = lane; p < T / 2; p += 64) {
                const int t = 2 * p;
                float ga[2], gb[2];
                if constexpr (!UP) {      // the gradient is already at the latent's resolution (polyphase backward-data)
                    typedef float f2g __attribute__((ext_vector_type(2)));
                    const f2g a2 = *(const f2g*)(g0 + t), b2 = *(const f2g*)(g1r + t);
                    ga[0] = a2[0], ga[1] = a2[1], gb[0] = b2[0], gb[1] = b2[1];
                } else if (p == 0 || p == T / 2 - 1) {
                    ga[0] = up2_adjoint(g0, T, t); ga[1] = up2_adjoint(g0, T, t + 1);
                    gb[0] = up2_adjoint(g1r, T, t); gb[1] = up2_adjoint(g1r, T, t + 1);
                } else {
                    const nef_f32x4 a4 = *(const nef_f32x4*)(g0 + 2 * t), b4 = *(const nef_f32x4*)(g1r + 2 * t);
                    const float al = g0[2 * t - 1], ar = g0[2 * t + 4], bl = g1r[2 * t - 1], br = g1r[2 * t + 4];
                    ga[0] = 0.25f * al + 0.75f * a4[0] + 0.75f * a4[1] + 0.25f * a4[2];
                    ga[1] = 0.25f * a4[1] + 0.75f * a4[2] + 0.75f * a4[3] + 0.25f * ar;
                    gb[0] = 0.25f * bl + 0.75f * b4[0] + 0.75f * b4[1] + 0.25f * b4[2];
                    gb[1] = 0.25f * b4[1] + 0.75f * b4[2] + 0.75f * b4[3] + 0.25f * br;
                }
                typedef float f2 __attribute__((ext_vector_type(2)));
                const f2 l2 = *(const f2*)(lat + t);
                const f2 pk2 = *(const f2*)(zsrc + (int64_t)pick_v * 128 * T + t);
#pragma unroll
                for (int e = 0; e < 2; ++e) acc += ga[e] * l2[e] + gb[e] * pk2[e];
                for (int v = 0; v < V; ++v) {
                    f2 o;
#pragma unroll
                    for (int e = 0; e < 2; ++e) o[e] = f * ga[e] / fv + (v == pick_v ? f * gb[e] : 0.f);
                    if (relu_z1 && first) {
                        const f2 z = *(const f2*)(zsrc + (int64_t)v * 128 * T + t);
                        if (!(z[0] > 0.f)) o[0] = 0.f;
                        if (!(z[1] > 0.f)) o[1] = 0.f;
                    }
                    *(f2*)(gdst + (int64_t)v * 128 * T + t) = o;
                }
            }
            acc = nef_wave_sum(acc);
            if (lane == 0) gq[row] = acc;
            continue;
        }
        for (int t = lane; t < T; t += 64) {
            float ga, gb, gc;
            if (UP) {
                ga = up2_adjoint(g0, T, t);
                gb = up2_adjoint(g0 + gpass, T, t);
                gc = SHARED ? 0.f : up2_adjoint(g0 + 2 * gpass, T, t);
            } else {
                ga = g0[t]; gb = g0[gpass + t]; gc = SHARED ? 0.f : g0[2 * gpass + t];
            }
            // pass that uses the picked lead for this half: D1 for the z1 half, D2 for the z2 half
            const float g_pick = SHARED ? gb : (first ? gb : gc);
            const float g_mean = SHARED ? ga : (first ? (ga + gc) : (ga + gb));
            const float l = lat[t];
            const float pk = zsrc[(int64_t)pick_v * 128 * T + t];
            acc += g_mean * l + g_pick * pk;
            const float gm = f * g_mean / fv;
            for (int v = 0; v < V; ++v) {
                float o = gm + (v == pick_v ? f * g_pick : 0.f);
                // relu_z1: z1 is a ReLU output (z1_conv's block): mask its gradient here instead of in a pass of its own
                if (relu_z1 && first && !(zsrc[(int64_t)v * 128 * T + t] > 0.f)) o = 0.f;
                gdst[(int64_t)v * 128 * T + t] = o;
            }
        }
        acc = nef_wave_sum(acc);
        if (lane == 0) gq[row] = acc;
    }
}

// mix_bwd_kernel<false, true> for T % 4 == 2 (configs[1]: T = 1250) on 16-byte accesses: a wave takes two consecutive channel rows
// (c even, c + 1), one aligned span of T / 2 four-float vectors in every tensor involved (see lead_mean_mix_shared_pair_kernel); the
// vector that straddles the rows carries both rows' scale and feeds both rows' gq sum.  gz1 / gz2r: the same expressions per element
// (bit-identical); gq: the same terms in another lane order, as between the two paths of mix_bwd_kernel.
__global__ __launch_bounds__(256) void mix_bwd_shared_pair_kernel(const float* __restrict__ gD, const float* __restrict__ latent,
                                                                  const float* __restrict__ z1, const float* __restrict__ z2r,
                                                                  const float* __restrict__ q, float* __restrict__ gz1,
                                                                  float* __restrict__ gz2r, float* __restrict__ gq, int B, int V, int T,
                                                                  int c1, int c2, const int32_t* __restrict__ choice_dev, int relu_z1) {
    typedef float vec __attribute__((ext_vector_type(4)));
    if (choice_dev) { c1 = choice_dev[0]; c2 = choice_dev[1]; }
    const int64_t pairs = (int64_t)B * 128;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float fv = (float)V;
    const int TW = T / 2, mid = T / 4;
    const int64_t gpass = pairs * TW;
    const int64_t lead = (int64_t)128 * T / 4;
    for (int64_t pr = (int64_t)blockIdx.x * 4 + wave; pr < pairs; pr += (int64_t)gridDim.x * 4) {
        const int64_t row = 2 * pr;
        const int b = (int)(row >> 8), c = (int)(row & 255);
        const bool first = c < 128;
        const int pick_v = first ? c1 : c2;
        const float f0 = q[row], f1 = q[row + 1];
        const vec* lat = (const vec*)latent + pr * TW;
        const vec* zsrc = (const vec*)((first ? z1 : z2r) + ((int64_t)b * V * 128 + (c & 127)) * T);
        vec* gdst = (vec*)((first ? gz1 : gz2r) + ((int64_t)b * V * 128 + (c & 127)) * T);
        const vec* g0 = (const vec*)gD + pr * TW;
        const vec* g1r = g0 + gpass;
        const bool mask = relu_z1 && first;
        float acc0 = 0.f, acc1 = 0.f;
        for (int t = lane; t < TW; t += 64) {
            const vec ga = g0[t], gb = g1r[t], l4 = lat[t], pk = zsrc[(int64_t)pick_v * lead + t];
            const vec fq = t < mid ? vec{f0, f0, f0, f0} : (t > mid ? vec{f1, f1, f1, f1} : vec{f0, f0, f1, f1});
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float term = ga[e] * l4[e] + gb[e] * pk[e];
                if (t < mid || (t == mid && e < 2)) acc0 += term; else acc1 += term;
            }
            for (int v = 0; v < V; ++v) {
                vec o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fq[e] * ga[e] / fv + (v == pick_v ? fq[e] * gb[e] : 0.f);
                if (mask) {
                    const vec z = zsrc[(int64_t)v * lead + t];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (!(z[e] > 0.f)) o[e] = 0.f;
                }
                gdst[(int64_t)v * lead + t] = o;
            }
        }
        acc0 = nef_wave_sum(acc0);
        acc1 = nef_wave_sum(acc1);
        if (lane == 0) { gq[row] = acc0; gq[row + 1] = acc1; }
    }
}

// ------------------------------------------------------------------------------------------------
// nn.Upsample(scale_factor=2, mode='linear', align_corners=False)        (:102,:104)
//   y[2m] = .25 x[m-1] + .75 x[m] (m>=1), y[0] = x[0];  y[2m+1] = .75 x[m] + .25 x[min(m+1,Tin-1)]
// written as torch's CPU kernel evaluates it: lambda0*x[i0] + lambda1*x[i1].
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float up2_at(const float* __restrict__ xr, int Tin, int i) {
    float src = 0.5f * ((float)i + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    int i0 = (int)src;
    if (i0 > Tin - 1) i0 = Tin - 1;
    const int i1 = i0 + (i0 < Tin - 1 ? 1 : 0);
    const float l1 = src - (float)i0;
    const float l0 = 1.f - l1;
    return l0 * xr[i0] + l1 * xr[i1];
}

// y = upsample2(relu(x*a[p][c] + b[p][c])): the BatchNorm affine + ReLU of the producing layer folded into the resample
__global__ void upsample2_aff_fwd_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                         const float* __restrict__ b, float* __restrict__ y, int64_t rows, int C, int Tin,
                                         int Bp) {
    const int To = 2 * Tin;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const int c = (int)(row % C);
        const int p = (int)((row / C) / Bp);
        const float af = a[p * C + c], bf = b[p * C + c];
        const float* xr = x + row * Tin;
        float* yr = y + row * To;
        for (int i = lane; i < To; i += 64) {
            float src = 0.5f * ((float)i + 0.5f) - 0.5f;
            if (src < 0.f) src = 0.f;
            int i0 = (int)src;
            if (i0 > Tin - 1) i0 = Tin - 1;
            const int i1 = i0 + (i0 < Tin - 1 ? 1 : 0);
            const float l1 = src - (float)i0;
            yr[i] = (1.f - l1) * fmaxf(fmaf(xr[i0], af, bf), 0.f) + l1 * fmaxf(fmaf(xr[i1], af, bf), 0.f);
        }
    }
}

__global__ void upsample2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t N, int Tin) {
    const int To = 2 * Tin;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < N; row += (int64_t)gridDim.x * 4) {
        const float* xr = x + row * Tin;
        float* yr = y + row * To;
        for (int i = lane; i < To; i += 64) yr[i] = up2_at(xr, Tin, i);
    }
}

// transpose of the above.  x[m] is read by outputs 2m-1 (weight .25), 2m (.75), 2m+1 (.75), 2m+2 (.25); the first and
// the last sample of a row also collect the clamped edge taps.  Lane m loads the pair (g[2m], g[2m+1]) as one 8-byte
// access (fully coalesced) and takes g[2m-1] / g[2m+2] from its neighbours with wave shuffles.
// One workgroup per row, four outputs per thread (Tin even): y[4q..4q+3] from x[2q-1..2q+2], same operation order as above.
__global__ __launch_bounds__(256) void upsample2_aff_fwd_rows(const float* __restrict__ x, const float* __restrict__ a,
                                                              const float* __restrict__ b, float* __restrict__ y, int C,
                                                              int Tin, int Bp) {
    const int64_t row = blockIdx.x;
    const int c = (int)(row % C);
    const int p = (int)((row / C) / Bp);
    const float af = a[p * C + c], bf = b[p * C + c];
    const float* xr = x + row * Tin;
    nef_f32x4* yr = (nef_f32x4*)(y + row * 2 * Tin);
    const int Q = Tin >> 1;
    for (int q = threadIdx.x; q < Q; q += 256) {
        const float2 m = *reinterpret_cast<const float2*>(xr + 2 * q);
        const float xl = xr[q > 0 ? 2 * q - 1 : 0], xh = xr[2 * q + 2 < Tin ? 2 * q + 2 : Tin - 1];
        const float r0 = fmaxf(fmaf(xl, af, bf), 0.f), r1 = fmaxf(fmaf(m.x, af, bf), 0.f);
        const float r2 = fmaxf(fmaf(m.y, af, bf), 0.f), r3 = fmaxf(fmaf(xh, af, bf), 0.f);
        nef_f32x4 o;
        // i = 4q: src = 2q - .25 (clamped to 0 at q = 0: l1 = 0, both taps are x[0])
        o[0] = q > 0 ? 0.25f * r0 + 0.75f * r1 : 1.f * r1 + 0.f * r2;
        o[1] = 0.75f * r1 + 0.25f * r2;
        o[2] = 0.25f * r1 + 0.75f * r2;
        o[3] = 0.75f * r2 + 0.25f * r3;
        yr[q] = o;
    }
}


__global__ void upsample2_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int64_t N, int Tin) {
    const int To = 2 * Tin;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < N; row += (int64_t)gridDim.x * 4) {
        const float* gr = gy + row * To;
        float* gxr = gx + row * Tin;
        for (int m0 = 0; m0 < Tin; m0 += 64) {
            const int m = m0 + lane;
            const bool in = m < Tin;
            float2 v = make_float2(0.f, 0.f);
            if (in) v = *reinterpret_cast<const float2*>(gr + 2 * m);
            float left = __shfl_up(v.y, 1, 64);       // g[2m-1] from lane-1
            float right = __shfl_down(v.x, 1, 64);    // g[2m+2] from lane+1
            if (!in) continue;
            if (lane == 0 && m > 0) left = gr[2 * m - 1];
            if (lane == 63 && m + 1 < Tin) right = gr[2 * m + 2];
            float s;
            if (m == 0 || m == Tin - 1) s = up2_bwd_edge(gr, Tin, m);
            else s = 0.25f * left + 0.75f * v.x + 0.75f * v.y + 0.25f * right;
            gxr[m] = s;
        }
    }
}

// One workgroup per row, two outputs per thread from one 16-byte load (Tin even)
__global__ __launch_bounds__(256) void upsample2_bwd_rows(const float* __restrict__ gy, float* __restrict__ gx, int Tin) {
    const int64_t row = blockIdx.x;
    const float* gr = gy + row * 2 * Tin;
    float2* gxr = reinterpret_cast<float2*>(gx + row * Tin);
    const int Q = Tin >> 1;
    for (int q = threadIdx.x; q < Q; q += 256) {
        const nef_f32x4 v = *(const nef_f32x4*)(gr + 4 * q);
        float2 o;
        if (q == 0) {
            o.x = up2_bwd_edge(gr, Tin, 0);
        } else {
            o.x = 0.25f * gr[4 * q - 1] + 0.75f * v[0] + 0.75f * v[1] + 0.25f * v[2];
        }
        if (q == Q - 1) {
            o.y = up2_bwd_edge(gr, Tin, Tin - 1);
        } else {
            o.y = 0.25f * v[1] + 0.75f * v[2] + 0.75f * v[3] + 0.25f * gr[4 * q + 4];
        }
        gxr[q] = o;
    }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm1d in training mode over P stacked passes                       (:19,:22 )
// ------------------------------------------------------------------------------------------------
constexpr int BN_SPLIT = 8;

// partial sums in double: part[(p*C + c)*BN_SPLIT + sp][2]
__global__ __launch_bounds__(256) void bn_stats_partial(const float* __restrict__ x, double* __restrict__ part, int P,
                                                        int Bp, int C, int L) {
    __shared__ double sm[4];
    int bid = blockIdx.x;
    const int sp = bid % BN_SPLIT;
    bid /= BN_SPLIT;
    const int c = bid % C;
    const int p = bid / C;
    double s1 = 0.0, s2 = 0.0;
    if ((L & 3) == 0) {
        // 16-byte loads, two rows in flight per trip: this pass is pure HBM streaming, latency hidden by loads in flight
        const int L4 = L >> 2;
        for (int b = sp; b < Bp; b += 2 * BN_SPLIT) {
            const nef_f32x4* r0 = (const nef_f32x4*)(x + (((int64_t)p * Bp + b) * C + c) * L);
            const bool two = b + BN_SPLIT < Bp;
            const nef_f32x4* r1 = two ? (const nef_f32x4*)(x + (((int64_t)p * Bp + b + BN_SPLIT) * C + c) * L) : r0;
            for (int t = threadIdx.x; t < L4; t += 256) {
                const nef_f32x4 u = r0[t];
                nef_f32x4 w = r1[t];
                if (!two) w = nef_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double a = (double)u[e], bq = (double)w[e];
                    s1 += a;
                    s2 += a * a;
                    s1 += bq;
                    s2 += bq * bq;
                }
            }
        }
    } else {
        for (int b = sp; b < Bp; b += BN_SPLIT) {
            const float* row = x + (((int64_t)p * Bp + b) * C + c) * L;
            for (int t = threadIdx.x; t < L; t += 256) {
                const double v = (double)row[t];
                s1 += v;
                s2 += v * v;
            }
        }
    }
    s1 = nef_block_sum_d(s1, sm);
    s2 = nef_block_sum_d(s2, sm);
    if (threadIdx.x == 0) {
        part[((int64_t)(p * C + c) * BN_SPLIT + sp) * 2] = s1;
        part[((int64_t)(p * C + c) * BN_SPLIT + sp) * 2 + 1] = s2;
    }
}

__global__ void bn_stats_final(const double* __restrict__ part, const float* __restrict__ gamma,
                               const float* __restrict__ beta, float* __restrict__ running_mean,
                               float* __restrict__ running_var, float* __restrict__ mean, float* __restrict__ invstd,
                               float* __restrict__ a, float* __restrict__ b, int P, int Bp, int C, int L, float eps,
                               float momentum, int nsplit, int64_t* __restrict__ nbt) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (nbt && c == 0) nbt[0] += P;      // BatchNorm1d.num_batches_tracked: one per pass (torch/nn/modules/batchnorm.py)
    const double n = (double)Bp * (double)L;
    float rm = running_mean ? running_mean[c] : 0.f;
    float rv = running_var ? running_var[c] : 1.f;
    for (int p = 0; p < P; ++p) {
        double s1 = 0.0, s2 = 0.0;
        for (int sp = 0; sp < nsplit; ++sp) {       // fixed order: deterministic
            s1 += part[((int64_t)(p * C + c) * nsplit + sp) * 2];
            s2 += part[((int64_t)(p * C + c) * nsplit + sp) * 2 + 1];
        }
        const double m = s1 / n;
        double var = s2 / n - m * m;
        if (var < 0.0) var = 0.0;
        const float mf = (float)m;
        const float is = (float)(1.0 / sqrt(var + (double)eps));
        mean[p * C + c] = mf;
        invstd[p * C + c] = is;
        const float af = gamma[c] * is;
        a[p * C + c] = af;
        b[p * C + c] = beta[c] - mf * af;
        const float unbiased = (float)(var * (n / (n - 1.0)));
        rm = (1.f - momentum) * rm + momentum * mf;
        rv = (1.f - momentum) * rv + momentum * unbiased;
    }
    if (running_mean) running_mean[c] = rm;
    if (running_var) running_var[c] = rv;
}

__global__ void bn_eval_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv,
                                      float* __restrict__ a, float* __restrict__ b, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float is = 1.0f / sqrtf(rv[c] + eps);
    const float af = gamma[c] * is;
    a[c] = af;
    b[c] = beta[c] - rm[c] * af;
}

// eval-mode BatchNorm folded into the preceding conv: w'[co][..] = a[co] * w[co][..], bias' = a * bias + b
__global__ void fold_bn_kernel(const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ a,
                               const float* __restrict__ b, float* __restrict__ w_out, float* __restrict__ bias_out,
                               int Cout, int inner) {
    const int64_t n = (int64_t)Cout * inner;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(i / inner);
        w_out[i] = a[co] * w[i];
        if (i % inner == 0) bias_out[co] = a[co] * bias[co] + b[co];
    }
}

__global__ void affine_relu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                       const float* __restrict__ b, float* __restrict__ y, int P, int Bp, int C,
                                       int L) {
    const int64_t rows = (int64_t)P * Bp * C;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const int c = (int)(row % C);
        const int p = (int)(row / ((int64_t)Bp * C));
        const float af = a[p * C + c], bf = b[p * C + c];
        const float* xr = x + row * L;
        float* yr = y + row * L;
        for (int t = lane; t < L; t += 64) yr[t] = fmaxf(fmaf(xr[t], af, bf), 0.f);
    }
}

// backward reductions per (pass, channel): s1 = sum g, s2 = sum g * xhat, g = gy * [x*a+b > 0]
// go = gout * out * (1 - out) / 3: the gradient at the last conv's pre-activation (sigmoid(x/3), model_nefnet.py:168)
__global__ void outconv_go_kernel(const float* __restrict__ gout, const float* __restrict__ out, float* __restrict__ go,
                                  int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float o = out[i];
        go[i] = gout[i] * (o * (1.f - o)) / 3.0f;
    }
}

// Four consecutive values of g[c][t] = w[c][0]*go[t+1] + w[c][1]*go[t] + w[c][2]*go[t-1] (the input gradient of the last
// conv, same expression as outconv_bwd_data_kernel) from one go row, so that BatchNorm-backward can read the small go
// instead of a materialised [N][64][L] gradient.
__device__ __forceinline__ nef_f32x4 oc_grad4(const float* __restrict__ gor, int t4, int L, float w0, float w1, float w2) {
    const nef_f32x4 m = *(const nef_f32x4*)(gor + 4 * t4);
    const float lo = t4 > 0 ? gor[4 * t4 - 1] : 0.f;
    const float hi = 4 * t4 + 4 < L ? gor[4 * t4 + 4] : 0.f;
    nef_f32x4 g;
    g[0] = w0 * m[1] + w1 * m[0] + w2 * lo;
    g[1] = w0 * m[2] + w1 * m[1] + w2 * m[0];
    g[2] = w0 * m[3] + w1 * m[2] + w2 * m[1];
    g[3] = w0 * hi + w1 * m[3] + w2 * m[2];
    return g;
}

// four consecutive values of the x2-upsampling adjoint (positions 4*t4 .. 4*t4+3 of an L-long row) from the 2L-long
// gradient row `gr`: the expressions of up2_adjoint / upsample2_bwd_rows
__device__ __forceinline__ nef_f32x4 up2_adjoint4(const float* __restrict__ gr, int t4, int L) {
    const nef_f32x4 v0 = *(const nef_f32x4*)(gr + 8 * t4), v1 = *(const nef_f32x4*)(gr + 8 * t4 + 4);
    const float lo = t4 > 0 ? gr[8 * t4 - 1] : 0.f;
    const float hi = 8 * t4 + 8 < 2 * L ? gr[8 * t4 + 8] : 0.f;
    nef_f32x4 g;
    g[0] = 0.25f * lo + 0.75f * v0[0] + 0.75f * v0[1] + 0.25f * v0[2];
    g[1] = 0.25f * v0[1] + 0.75f * v0[2] + 0.75f * v0[3] + 0.25f * v1[0];
    g[2] = 0.25f * v0[3] + 0.75f * v1[0] + 0.75f * v1[1] + 0.25f * v1[2];
    g[3] = 0.25f * v1[1] + 0.75f * v1[2] + 0.75f * v1[3] + 0.25f * hi;
    if (t4 == 0) g[0] = up2_bwd_edge(gr, L, 0);
    if (4 * t4 + 3 == L - 1) g[3] = up2_bwd_edge(gr, L, L - 1);
    return g;
}

// MODE 1: `gy` is the go tensor [P*Bp][L] of the last conv and `ocw` its weight [C][3]; MODE 2: `gy` is the gradient wrt
// the x2-UPSAMPLED activation [rows][2L]; in both cases g is rebuilt on the fly (L % 4 == 0)
template <int MODE>
__global__ __launch_bounds__(256) void bn_bwd_partial(const float* __restrict__ gy, const float* __restrict__ x,
                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      const float* __restrict__ a, const float* __restrict__ b,
                                                      double* __restrict__ part, int P, int Bp, int C, int L,
                                                      const float* __restrict__ ocw) {
    __shared__ double sm[4];
    int bid = blockIdx.x;
    const int sp = bid % BN_SPLIT;
    bid /= BN_SPLIT;
    const int c = bid % C;
    const int p = bid / C;
    const float mf = mean[p * C + c], is = invstd[p * C + c], af = a[p * C + c], bf = b[p * C + c];
    constexpr bool OC = MODE == 1, UPG = MODE == 2;
    const float w0 = OC ? ocw[c * 3] : 0.f, w1 = OC ? ocw[c * 3 + 1] : 0.f, w2 = OC ? ocw[c * 3 + 2] : 0.f;
    double s1 = 0.0, s2 = 0.0;
    // accumulate in double from the first element: sum(g) cancels heavily and k1 = sum(g)/n shifts every gx
    if ((L & 3) == 0) {
        const int L4 = L >> 2;
        for (int bb = sp; bb < Bp; bb += BN_SPLIT) {
            const int64_t off = (((int64_t)p * Bp + bb) * C + c) * L;
            const nef_f32x4* xr = (const nef_f32x4*)(x + off);
            const nef_f32x4* gr = (const nef_f32x4*)(gy + off);
            const float* gor = gy + ((int64_t)p * Bp + bb) * L;
            const float* gur = gy + 2 * off;
#pragma unroll 2
            for (int t = threadIdx.x; t < L4; t += 256) {
                const nef_f32x4 xv = xr[t],
                                gv = OC ? oc_grad4(gor, t, L, w0, w1, w2) : (UPG ? up2_adjoint4(gur, t, L) : gr[t]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = fmaf(xv[e], af, bf) > 0.f ? gv[e] : 0.f;
                    s1 += (double)g;
                    s2 += (double)g * (double)((xv[e] - mf) * is);
                }
            }
        }
    } else {
        for (int bb = sp; bb < Bp; bb += BN_SPLIT) {
            const int64_t off = (((int64_t)p * Bp + bb) * C + c) * L;
            for (int t = threadIdx.x; t < L; t += 256) {
                const float xv = x[off + t];
                const float g = fmaf(xv, af, bf) > 0.f ? gy[off + t] : 0.f;
                s1 += (double)g;
                s2 += (double)g * (double)((xv - mf) * is);
            }
        }
    }
    s1 = nef_block_sum_d(s1, sm);
    s2 = nef_block_sum_d(s2, sm);
    if (threadIdx.x == 0) {
        part[((int64_t)(p * C + c) * BN_SPLIT + sp) * 2] = s1;
        part[((int64_t)(p * C + c) * BN_SPLIT + sp) * 2 + 1] = s2;
    }
}

// coef[(p*C+c)*2] = s1/n, s2/n ; ggamma[c] = sum_p s2 ; gbeta[c] = sum_p s1
__global__ void bn_bwd_final(const double* __restrict__ part, float* __restrict__ coef, float* __restrict__ ggamma,
                             float* __restrict__ gbeta, int P, int Bp, int C, int L, int nsplit = BN_SPLIT) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double n = (double)Bp * (double)L;
    double g1 = 0.0, g2 = 0.0;
    for (int p = 0; p < P; ++p) {
        double s1 = 0.0, s2 = 0.0;
        for (int sp = 0; sp < nsplit; ++sp) {
            s1 += part[((int64_t)(p * C + c) * nsplit + sp) * 2];
            s2 += part[((int64_t)(p * C + c) * nsplit + sp) * 2 + 1];
        }
        coef[(p * C + c) * 2] = (float)(s1 / n);
        coef[(p * C + c) * 2 + 1] = (float)(s2 / n);
        g1 += s1;
        g2 += s2;
    }
    gbeta[c] = (float)g1;
    ggamma[c] = (float)g2;
}

__global__ void bn_bwd_apply(const float* __restrict__ gy, const float* __restrict__ x, const float* __restrict__ mean,
                             const float* __restrict__ invstd, const float* __restrict__ a, const float* __restrict__ b,
                             const float* __restrict__ coef, float* __restrict__ gx, double* __restrict__ rowsum, int P,
                             int Bp, int C, int L) {
    const int64_t rows = (int64_t)P * Bp * C;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const int c = (int)(row % C);
        const int p = (int)(row / ((int64_t)Bp * C));
        const int pc = p * C + c;
        const float mf = mean[pc], is = invstd[pc], af = a[pc], bf = b[pc];
        const float k1 = coef[pc * 2], k2 = coef[pc * 2 + 1];
        const float* xr = x + row * L;
        const float* gr = gy + row * L;
        float* gxr = gx + row * L;
        double rs = 0.0;
        if ((L & 3) == 0) {
            const int L4 = L >> 2;
#pragma unroll 2
            for (int t = lane; t < L4; t += 64) {
                const nef_f32x4 xv = ((const nef_f32x4*)xr)[t], gv = ((const nef_f32x4*)gr)[t];
                nef_f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = fmaf(xv[e], af, bf) > 0.f ? gv[e] : 0.f;
                    o[e] = af * (g - k1 - (xv[e] - mf) * is * k2);
                    rs += (double)o[e];
                }
                ((nef_f32x4*)gxr)[t] = o;
            }
        } else {
            for (int t = lane; t < L; t += 64) {
                const float xv = xr[t];
                const float g = fmaf(xv, af, bf) > 0.f ? gr[t] : 0.f;
                const float o = af * (g - k1 - (xv - mf) * is * k2);
                gxr[t] = o;
                rs += (double)o;
            }
        }
        if (rowsum) {                      // per-row sum of gx: the bias gradient of the conv feeding this BN
            rs = nef_wave_sum_d(rs);
            if (lane == 0) rowsum[row] = rs;
        }
    }
}

// Same pass, one workgroup per (pass, sample, channel) row with 16-byte accesses (L % 4 == 0): the row's six constants
// are wave-uniform scalar loads, every thread has all its loads in flight at once, and the optional row sum costs one
// block reduction per row instead of a double-precision shuffle tree per wave.
template <int MODE>
__global__ __launch_bounds__(256) void bn_bwd_apply_rows(const float* __restrict__ gy, const float* __restrict__ x,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ coef, float* __restrict__ gx,
                                                         double* __restrict__ rowsum, int Bp, int C, int L4,
                                                         const float* __restrict__ ocw, int deint = 0) {
    // deint: the row is written PHASE-MAJOR -- gx[row][t & 1][t >> 1], i.e. as the two half-length rows 2 row, 2 row + 1 of a
    // [.., 2 C, L / 2] tensor: the operand layout of the polyphase backward passes through a x2 upsampling
    __shared__ double sm[4];
    const int64_t row = blockIdx.x;
    const int c = (int)(row % C);
    const int p = (int)(row / ((int64_t)Bp * C));
    const int pc = p * C + c;
    const float mf = mean[pc], is = invstd[pc], af = a[pc], bf = b[pc];
    const float k1 = coef[pc * 2], k2 = coef[pc * 2 + 1];
    constexpr bool OC = MODE == 1, UPG = MODE == 2;
    const float w0 = OC ? ocw[c * 3] : 0.f, w1 = OC ? ocw[c * 3 + 1] : 0.f, w2 = OC ? ocw[c * 3 + 2] : 0.f;
    const nef_f32x4* xr = (const nef_f32x4*)(x + row * 4 * L4);
    const nef_f32x4* gr = (const nef_f32x4*)(gy + row * 4 * L4);
    const float* gor = gy + (row / C) * 4 * L4;          // OC: the sample's go row
    const float* gur = gy + row * 8 * L4;                // UPG: this row at twice the length
    nef_f32x4* gxr = (nef_f32x4*)(gx + row * 4 * L4);
    double rs = 0.0;
    for (int t0 = 0; t0 < L4; t0 += 1024) {
        nef_f32x4 xv[4], gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + u * 256 + threadIdx.x;
            if (t < L4) {
                xv[u] = xr[t];
                gv[u] = OC ? oc_grad4(gor, t, 4 * L4, w0, w1, w2) : (UPG ? up2_adjoint4(gur, t, 4 * L4) : gr[t]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + u * 256 + threadIdx.x;
            if (t < L4) {
                nef_f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = fmaf(xv[u][e], af, bf) > 0.f ? gv[u][e] : 0.f;
                    o[e] = af * (g - k1 - (xv[u][e] - mf) * is * k2);
                    rs += (double)o[e];
                }
                if (deint) {
                    float* const gb_ = gx + row * 4 * L4;
                    ((nef_f32x2*)gb_)[t] = nef_f32x2{o[0], o[2]};
                    ((nef_f32x2*)(gb_ + 2 * L4))[t] = nef_f32x2{o[1], o[3]};
                } else
                gxr[t] = o;
            }
        }
    }
    if (rowsum) {
        rs = nef_block_sum_d(rs, sm);
        if (threadIdx.x == 0) rowsum[row] = rs;
    }
}

// BatchNorm-backward apply of the FIRST decoder BatchNorm with the pass adjoint of pass_combine_bwd fused in: one
// workgroup per (sample, channel) computes the three passes' gx and writes the four half-conv output gradients
// (A[mean] = g0 + g2, A[pick] = g1, B[mean] = g0 + g1, B[pick] = g2) -- gx itself is only needed for its row sums.
__global__ __launch_bounds__(256) void bn_bwd_apply_combine3(const float* __restrict__ gy, const float* __restrict__ x,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, const float* __restrict__ a,
                                                             const float* __restrict__ b, const float* __restrict__ coef,
                                                             float* __restrict__ gP2, double* __restrict__ rowsum, int Bp,
                                                             int C, int L, int deint = 0) {
    __shared__ double sm[4];
    const int bb = blockIdx.x / C, c = blockIdx.x % C;
    float mf[3], is[3], af[3], bf[3], k1[3], k2[3];
    const float* xr[3];
    const float* gr[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int pc = p * C + c;
        mf[p] = mean[pc]; is[p] = invstd[pc]; af[p] = a[pc]; bf[p] = b[pc];
        k1[p] = coef[pc * 2]; k2[p] = coef[pc * 2 + 1];
        const int64_t off = (((int64_t)p * Bp + bb) * C + c) * L;
        xr[p] = x + off;
        gr[p] = gy + off;
    }
    float* am = gP2 + ((int64_t)bb * 2 * C + c) * L;
    float* bm = gP2 + ((int64_t)bb * 2 * C + C + c) * L;
    float* ap = gP2 + ((int64_t)(Bp + bb) * 2 * C + c) * L;
    float* bp = gP2 + ((int64_t)(Bp + bb) * 2 * C + C + c) * L;
    double rs[3] = {0.0, 0.0, 0.0};
    for (int t = threadIdx.x; t < L; t += 256) {
        float o[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const float xv = xr[p][t];
            const float g = fmaf(xv, af[p], bf[p]) > 0.f ? gr[p][t] : 0.f;
            o[p] = af[p] * (g - k1[p] - (xv - mf[p]) * is[p] * k2[p]);
            rs[p] += (double)o[p];
        }
        const int tq = deint ? (t & 1) * (L >> 1) + (t >> 1) : t;      // (phase-major rows, see bn_bwd_apply_rows)
        am[tq] = o[0] + o[2];
        ap[tq] = o[1];
        bm[tq] = o[0] + o[1];
        bp[tq] = o[2];
    }
    if (rowsum) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const double v = nef_block_sum_d(rs[p], sm);
            if (threadIdx.x == 0) rowsum[((int64_t)p * Bp + bb) * C + c] = v;
        }
    }
}

// out[c] = sum over rows (p, b) of rowsum[(p*Bp + b)*C + c]; one workgroup per channel, fixed reduction tree
__global__ __launch_bounds__(256) void rowsum_to_channel(const double* __restrict__ rowsum, float* __restrict__ out,
                                                         int NB, int C) {
    __shared__ double sm[4];
    const int c = blockIdx.x;
    double s = 0.0;
    for (int n = threadIdx.x; n < NB; n += 256) s += rowsum[(int64_t)n * C + c];
    s = nef_block_sum_d(s, sm);
    if (threadIdx.x == 0) out[c] = (float)s;
}

// ------------------------------------------------------------------------------------------------
// Conv1d(C->1, k3, p1, bias) + sigmoid(x/3)                               (:106, :168)
// ------------------------------------------------------------------------------------------------
constexpr int OC_MAXC = 64;

template <bool VEC4>
__global__ __launch_bounds__(256) void outconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          int N, int C, int L, int tiles, const float* __restrict__ pa,
                                                          const float* __restrict__ pb, int Bp) {
    __shared__ float wl[OC_MAXC * 3];
    __shared__ float al[OC_MAXC], bl[OC_MAXC];
    const int n = blockIdx.x / tiles;
    for (int i = threadIdx.x; i < C * 3; i += 256) wl[i] = w[i];
    for (int i = threadIdx.x; i < C; i += 256) {
        al[i] = pa ? pa[(n / Bp) * C + i] : 1.f;
        bl[i] = pa ? pb[(n / Bp) * C + i] : 0.f;
    }
    __syncthreads();
    if (VEC4) {
        // four consecutive outputs per thread: one 16-byte load + the two neighbours per channel
        const int t = ((blockIdx.x % tiles) * 256 + threadIdx.x) * 4;
        if (t >= L) return;
        const float* xs = x + (int64_t)n * C * L + t;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const bool hl = t > 0, hr = t + 4 < L;
        for (int c = 0; c < C; ++c) {
            const float* r = xs + (int64_t)c * L;
            const nef_f32x4 m = *(const nef_f32x4*)r;
            float v[6] = {hl ? r[-1] : 0.f, m[0], m[1], m[2], m[3], hr ? r[4] : 0.f};
            if (pa) {   // BatchNorm affine + ReLU of the producing layer; padding stays zero
#pragma unroll
                for (int e = 0; e < 6; ++e) v[e] = fmaxf(fmaf(v[e], al[c], bl[c]), 0.f);
                if (!hl) v[0] = 0.f;
                if (!hr) v[5] = 0.f;
            }
            const float w0 = wl[c * 3], w1 = wl[c * 3 + 1], w2 = wl[c * 3 + 2];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[e] = fmaf(w0, v[e], acc[e]);
                acc[e] = fmaf(w1, v[e + 1], acc[e]);
                acc[e] = fmaf(w2, v[e + 2], acc[e]);
            }
        }
        nef_f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = 1.0f / (1.0f + expf(-((acc[e] + bias[0]) / 3.0f)));
        *(nef_f32x4*)(out + (int64_t)n * L + t) = o;
        return;
    }
    const int t = (blockIdx.x % tiles) * 256 + threadIdx.x;
    if (t >= L) return;
    const float* xs = x + (int64_t)n * C * L + t;
    float acc = 0.f;
    const bool hl = t > 0, hr = t < L - 1;
    for (int c = 0; c < C; ++c) {
        const float* r = xs + (int64_t)c * L;
        float xm = hl ? r[-1] : 0.f;
        float x0 = r[0];
        float xp = hr ? r[1] : 0.f;
        if (pa) {       // BatchNorm affine + ReLU of the producing layer; padding stays zero
            xm = hl ? fmaxf(fmaf(xm, al[c], bl[c]), 0.f) : 0.f;
            x0 = fmaxf(fmaf(x0, al[c], bl[c]), 0.f);
            xp = hr ? fmaxf(fmaf(xp, al[c], bl[c]), 0.f) : 0.f;
        }
        acc = fmaf(wl[c * 3], xm, acc);
        acc = fmaf(wl[c * 3 + 1], x0, acc);
        acc = fmaf(wl[c * 3 + 2], xp, acc);
    }
    acc += bias[0];
    const float z = acc / 3.0f;
    out[(int64_t)n * L + t] = 1.0f / (1.0f + expf(-z));
}

// go = gout * out * (1 - out) / 3 ; gx[n][c][t] = sum_k w[c][k] * go[n][t - k + 1]
__global__ __launch_bounds__(256) void outconv_bwd_data_kernel(const float* __restrict__ gout,
                                                               const float* __restrict__ out,
                                                               const float* __restrict__ w, float* __restrict__ gx,
                                                               int N, int C, int L, int tiles) {
    __shared__ float wl[OC_MAXC * 3];
    for (int i = threadIdx.x; i < C * 3; i += 256) wl[i] = w[i];
    __syncthreads();
    const int n = blockIdx.x / tiles;
    const int t = (blockIdx.x % tiles) * 256 + threadIdx.x;
    if (t >= L) return;
    const int64_t base = (int64_t)n * L;
    auto go_at = [&](int tt) -> float {
        if (tt < 0 || tt >= L) return 0.f;
        const float o = out[base + tt];
        return gout[base + tt] * (o * (1.f - o)) / 3.0f;
    };
    const float gm = go_at(t - 1), g0 = go_at(t), gp = go_at(t + 1);
    float* dst = gx + (int64_t)n * C * L + t;
    for (int c = 0; c < C; ++c)
        dst[(int64_t)c * L] = wl[c * 3] * gp + wl[c * 3 + 1] * g0 + wl[c * 3 + 2] * gm;
}

// gw[c][k] = sum_{n,t} go[n][t] * x[n][c][t+k-1] ; gb = sum go, with go = gout * out(1-out)/3.
// A workgroup walks (sample, 1024-sample tile) units: go of the tile (+1 halo each side) is built once in LDS, every x
// element is read once (coalesced) and feeds the three taps; per-channel sums stay in registers across units and are
// wave-reduced once.  OC_BLOCKS partial rows are then summed in a fixed order.
constexpr int OC_BLOCKS = 1024;
constexpr int OC_TILE = 1024;
constexpr int OC_CPW = OC_MAXC / 4;     // channels per wave

__global__ __launch_bounds__(256) void outconv_bwd_weight_partial(const float* __restrict__ gout,
                                                                  const float* __restrict__ out,
                                                                  const float* __restrict__ x, double* __restrict__ part,
                                                                  int N, int C, int L, int tiles,
                                                                  const float* __restrict__ pa,
                                                                  const float* __restrict__ pb, int Bp) {
    __shared__ float gol[OC_TILE + 2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc[OC_CPW][3];
#pragma unroll
    for (int j = 0; j < OC_CPW; ++j) acc[j][0] = acc[j][1] = acc[j][2] = 0.f;
    float bsum = 0.f;
    const int n_units = N * tiles;
    for (int unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        const int n = unit / tiles;
        const int t0 = (unit - n * tiles) * OC_TILE;
        __syncthreads();
        for (int i = threadIdx.x; i < OC_TILE + 2; i += 256) {
            const int t = t0 - 1 + i;
            float v = 0.f;
            if (t >= 0 && t < L) {
                const float o = out[(int64_t)n * L + t];
                v = gout[(int64_t)n * L + t] * (o * (1.f - o)) / 3.0f;
            }
            gol[i] = v;
        }
        __syncthreads();
        if ((L & 3) == 0) {
            // this lane's go values (4 runs of 4 positions + 1 halo each side) live in registers for all channels
            float gr[4][6];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 6; ++e) gr[i][e] = gol[(lane + 64 * i) * 4 + e];
#pragma unroll
            for (int j = 0; j < OC_CPW; ++j) {
                const int c = wave * OC_CPW + j;
                if (c < C) {
                    const float* xr = x + ((int64_t)n * C + c) * L + t0;
                    const float af = pa ? pa[(n / Bp) * C + c] : 1.f, bf = pa ? pb[(n / Bp) * C + c] : 0.f;
                    nef_f32x4 xv[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int tl = (lane + 64 * i) * 4;
                        xv[i] = t0 + tl < L ? *(const nef_f32x4*)(xr + tl) : nef_f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (t0 + (lane + 64 * i) * 4 < L) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float v = xv[i][e];
                                if (pa) v = fmaxf(fmaf(v, af, bf), 0.f);
                                acc[j][0] = fmaf(gr[i][e + 2], v, acc[j][0]);
                                acc[j][1] = fmaf(gr[i][e + 1], v, acc[j][1]);
                                acc[j][2] = fmaf(gr[i][e], v, acc[j][2]);
                            }
                        }
                    }
                }
            }
        } else
#pragma unroll
        for (int j = 0; j < OC_CPW; ++j) {
            const int c = wave * OC_CPW + j;
            if (c < C) {
                const float* xr = x + ((int64_t)n * C + c) * L + t0;
                const float af = pa ? pa[(n / Bp) * C + c] : 1.f, bf = pa ? pb[(n / Bp) * C + c] : 0.f;
#pragma unroll 4
                for (int tl = lane; tl < OC_TILE; tl += 64) {
                    if (t0 + tl < L) {
                        float xv = xr[tl];
                        if (pa) xv = fmaxf(fmaf(xv, af, bf), 0.f);
                        acc[j][0] = fmaf(gol[tl + 2], xv, acc[j][0]);
                        acc[j][1] = fmaf(gol[tl + 1], xv, acc[j][1]);
                        acc[j][2] = fmaf(gol[tl], xv, acc[j][2]);
                    }
                }
            }
        }
        if (wave == 0)
            for (int tl = lane; tl < OC_TILE; tl += 64) bsum += gol[tl + 1];
    }
    double* row = part + (int64_t)blockIdx.x * (C + 1) * 3;
#pragma unroll
    for (int j = 0; j < OC_CPW; ++j) {
        const int c = wave * OC_CPW + j;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float s = nef_wave_sum(acc[j][k]);
            if (lane == 0 && c < C) row[c * 3 + k] = (double)s;
        }
    }
    if (wave == 0) {
        const float s = nef_wave_sum(bsum);
        if (lane == 0) { row[C * 3] = 0.0; row[C * 3 + 1] = (double)s; row[C * 3 + 2] = 0.0; }
    }
}

// one workgroup per output value: sum the OC_BLOCKS partial rows
__global__ __launch_bounds__(256) void outconv_bwd_weight_final(const double* __restrict__ part, float* __restrict__ gw,
                                                                float* __restrict__ gb, int C, int nblk) {
    __shared__ double sm[4];
    const int i = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) s += part[(int64_t)b * (C + 1) * 3 + i];
    s = nef_block_sum_d(s, sm);
    if (threadIdx.x != 0) return;
    const int c = i / 3, k = i % 3;
    if (c < C) gw[c * 3 + k] = (float)s;
    else if (k == 1) gb[0] = (float)s;
}

// ------------------------------------------------------------------------------------------------
// losswrapper                                                    (codes/network/loss/losses.py:21-50)
// ------------------------------------------------------------------------------------------------
constexpr int LOSS_BLOCKS = 256;

__global__ __launch_bounds__(256) void loss_partial(const float* __restrict__ pred, const float* __restrict__ pp,
                                                    const float* __restrict__ pl, const float* __restrict__ tgt,
                                                    double* __restrict__ part, int64_t n, int reg_l2) {
    __shared__ double sm[4];
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float o = pred[i];
        s1 += (double)fabsf(o - pp[i]);
        s2 += (double)fabsf(o - pl[i]);
        const float d = o - tgt[i];
        s3 += (double)(reg_l2 ? d * d : fabsf(d));
    }
    s1 = nef_block_sum_d(s1, sm);
    s2 = nef_block_sum_d(s2, sm);
    s3 = nef_block_sum_d(s3, sm);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 3] = s1;
        part[blockIdx.x * 3 + 1] = s2;
        part[blockIdx.x * 3 + 2] = s3;
    }
}

__global__ __launch_bounds__(256) void loss_final(const double* __restrict__ part, float* __restrict__ losses, int nblk,
                                                  int64_t n, float f0, float f1, float f2, int use_mask) {
    __shared__ double sm[4];
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) {
        s1 += part[i * 3];
        s2 += part[i * 3 + 1];
        s3 += part[i * 3 + 2];
    }
    s1 = nef_block_sum_d(s1, sm);
    s2 = nef_block_sum_d(s2, sm);
    s3 = nef_block_sum_d(s3, sm);
    if (threadIdx.x != 0) return;
    const float l1 = (use_mask & 1) ? (float)(s1 / (double)n) : 0.f;
    const float l2 = (use_mask & 2) ? (float)(s2 / (double)n) : 0.f;
    const float l3 = (use_mask & 4) ? (float)(s3 / (double)n) : 0.f;
    const float t1 = l1 * f0, t2 = l2 * f1, t3 = l3 * f2;
    losses[0] = t1 + t2 + t3;
    losses[1] = t1;
    losses[2] = t2;
    losses[3] = t3;
}

__device__ __forceinline__ float sgnf(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

__global__ void loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ pp,
                                const float* __restrict__ pl, const float* __restrict__ tgt,
                                const float* __restrict__ gscale, float* __restrict__ g_pred, float* __restrict__ g_p,
                                float* __restrict__ g_l, int64_t n, float f0, float f1, float f2, int reg_l2,
                                int use_mask) {
    const float gs = gscale ? gscale[0] : 1.f;
    const float inv_n = 1.0f / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float o = pred[i];
        // d|sg(o) - p| / dp = sign(p - o)
        g_p[i] = (use_mask & 1) ? gs * f0 * inv_n * sgnf(pp[i] - o) : 0.f;
        g_l[i] = (use_mask & 2) ? gs * f1 * inv_n * sgnf(pl[i] - o) : 0.f;
        const float d = o - tgt[i];
        const float g3 = reg_l2 ? 2.f * d : sgnf(d);
        g_pred[i] = (use_mask & 4) ? gs * f2 * inv_n * g3 : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// torch.optim.SGD(momentum)                                 (codes/solver/optim_scheduler.py:10)
// ------------------------------------------------------------------------------------------------
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, int64_t n,
                           float lr, float mu, float gscale, int first, const float* __restrict__ skip, int32_t* skipped,
                           const float* __restrict__ lr_dev) {
    if (lr_dev) lr = lr_dev[0];      // a captured launch freezes its scalar arguments: the learning rate of a replayed step lives in memory
    if (skip && skip[0] > 0.f) {      // a tainted step (nef_h2_taint): parameters and momentum stay as they are
        if (skipped && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(skipped, 1);
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gv = g[i] * gscale;
        const float bv = first ? gv : fmaf(mu, buf[i], gv);
        buf[i] = bv;
        p[i] = p[i] - lr * bv;
    }
}

// ------------------------------------------------------------------------------------------------
// Round 6: the last torch elementwise kernels of the train step, as one launch each.
// amax_roll: ops.amax_roll's follow-up rule on the split-fp16 site table (was ~10 ATen launches: compares, ors, where, fill).
__global__ __launch_bounds__(256) void amax_roll_kernel(float* __restrict__ cur, float* __restrict__ nxt, int n, float up, float down,
                                                        int follow) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float c = cur[i], x = nxt[i];
    const bool upd = x > 0.f && (follow || c <= 0.f || x > up * c || x * down < c);
    if (upd) cur[i] = x;
    nxt[i] = 0.f;
}

// flatten: out[off_k .. off_k + n_k) = src_k for up to FLAT_MAX tensors per launch (the flat gradient buffer; was torch.cat)
constexpr int FLAT_MAX = 64;
struct FlatDesc { const float* src; int64_t n, off; };
struct FlatTable { FlatDesc d[FLAT_MAX]; };
__global__ __launch_bounds__(256) void flatten_kernel(FlatTable t, float* __restrict__ out) {
    const FlatDesc& d = t.d[blockIdx.y];
    const float* __restrict__ src = d.src;
    float* __restrict__ dst = out + d.off;
    const int64_t n = d.n;
    const int64_t stride = (int64_t)gridDim.x * 256;
    // 16-byte body when both sides are 16-byte aligned (every weight gradient of the step is), scalar otherwise
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        const int64_t n4 = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
            reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
        for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
    }
}

// regroup_halves: w [Co][2 Cih][K] <-> grouped [2 Co][Cih][K] (group = input-channel half; engine._regroup_halves / _ungroup_halves)
__global__ __launch_bounds__(256) void regroup_halves_kernel(const float* __restrict__ src, float* __restrict__ dst, int Co, int Cih,
                                                            int K, int inverse) {
    const int64_t n = (int64_t)2 * Co * Cih * K;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        // i indexes the GROUPED tensor [h][co][ci][k]
        const int k = (int)(i % K);
        const int64_t q = i / K;
        const int ci = (int)(q % Cih);
        const int64_t q2 = q / Cih;
        const int co = (int)(q2 % Co), h = (int)(q2 / Co);
        const int64_t j = (((int64_t)co * 2 + h) * Cih + ci) * K + k;      // the same element in [co][h][ci][k]
        if (inverse) dst[j] = src[i];
        else dst[i] = src[j];
    }
}

// step_words: the per-step host decisions of a replayed train step (the two Standin lead choices, the dropout seed) written into
// their device words by a launch that carries them as kernel ARGUMENTS -- stream-ordered and asynchronous, where the blocking
// host-to-device copies of rounds 1-5 made the host wait for the previous step before it could stage the next.
__global__ void step_words_kernel(int32_t* __restrict__ choice, int64_t* __restrict__ seed, int c1, int c2, int64_t sd) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        choice[0] = c1;
        choice[1] = c2;
        seed[0] = sd;
    }
}

// Diagnostics (bench.py --dry-collective): occupy `wgs` workgroups for `ticks` of the constant-rate wall clock -- a stand-in for a
// collective of modelled duration on the communication stream of a ONE-GPU run (the all-reduce it replaces would hold a few CUs
// of channel kernels for that long).  Bounded: the host side caps the duration at 50 ms.
__global__ __launch_bounds__(64) void debug_spin_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// ------------------------------------------------------------------------------------------------
// Polyphase form of conv1d(upsample2(x), w), K = 3.  Output 2 m + p of that conv is a K = 3 conv of the
// HALF-resolution x with the phase weights W'_p (taps on x[m - 1], x[m], x[m + 1]):
//     W'_0 = (0.75 w0 + 0.25 w1,  0.25 w0 + 0.75 w1 + 0.75 w2,  0.25 w2)
//     W'_1 = (0.25 w0,  0.75 w0 + 0.75 w1 + 0.25 w2,  0.25 w1 + 0.75 w2)
// (nn.Upsample(scale_factor=2, mode='linear'): u[2j] = 0.25 x[j-1] + 0.75 x[j], u[2j+1] = 0.75 x[j] + 0.25 x[j+1]).
// wsyn [R * 2][Cig][3], row 2 r + p for row r of w [R][Cig][3].  One thread per (r, ci).
// `Cr` > 0: TILE order for the polyphase forward launch (conv_h2_kernel PF) -- w has Cr rows per group, phase p of channel co of
// group g is row g 2 Cr + (co / 64) 128 + ((co / 32) & 1) 64 + p 32 + co % 32.
__global__ __launch_bounds__(256) void poly_weights_kernel(const float* __restrict__ w, float* __restrict__ ws, int R, int Cig, int Cr) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)R * Cig) return;
    const int r = (int)(i / Cig), ci = (int)(i - (int64_t)r * Cig);
    const float w0 = w[3 * i], w1 = w[3 * i + 1], w2 = w[3 * i + 2];
    int r0 = 2 * r, r1 = 2 * r + 1;
    if (Cr > 0) {
        const int g = r / Cr, co = r - g * Cr;
        r0 = g * 2 * Cr + (co >> 6) * 128 + ((co >> 5) & 1) * 64 + (co & 31);
        r1 = r0 + 32;
    }
    float* const o0 = ws + ((int64_t)r0 * Cig + ci) * 3;
    float* const o1 = ws + ((int64_t)r1 * Cig + ci) * 3;
    o0[0] = fmaf(0.75f, w0, 0.25f * w1);
    o0[1] = fmaf(0.25f, w0, 0.75f * (w1 + w2));
    o0[2] = 0.25f * w2;
    o1[0] = 0.25f * w0;
    o1[1] = fmaf(0.25f, w2, 0.75f * (w0 + w1));
    o1[2] = fmaf(0.75f, w2, 0.25f * w1);
}

// Row ends of the polyphase forward pass: the phase convs see x[-1] = x[0] and x[Tin] = x[Tin-1] (nn.Upsample's clamped sources)
// and therefore u[-1] = x[0], u[T] = x[Tin-1] where the conv's zero padding has 0:
//     y[c][0] -= sum_ci w[c][ci][0] x'[ci][0]        y[c][T-1] -= sum_ci w[c][ci][2] x'[ci][Tin-1]        (x' = the prologue's output)
// and the two columns' change in the BatchNorm statistics (sum, sum of squares) goes into slot 0 of the sample.  One workgroup per
// (sample, group).
__global__ __launch_bounds__(256) void poly_fwd_edge_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                            int B, int G, int Cr, int Cig, int T, const float* __restrict__ pa,
                                                            const float* __restrict__ pb, int Bp, float* __restrict__ slots, int nslot,
                                                            float* __restrict__ xedge) {
    extern __shared__ float xl[];      // [2][Cig]: x'[:, 0], x'[:, Tin - 1]
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const int Tin = T >> 1;
    const float* const xb = x + ((int64_t)b * G + g) * Cig * Tin;
    for (int i = threadIdx.x; i < 2 * Cig; i += 256) {
        const int ci = i >> 1, k = i & 1;
        float v = xb[(int64_t)ci * Tin + (k ? Tin - 1 : 0)];
        if (pa) {
            const int64_t pr = (int64_t)(b / Bp) * G * Cig + (int64_t)g * Cig + ci;
            v = fmaxf(fmaf(v, pa[pr], pb[pr]), 0.f);
        }
        xl[k * Cig + ci] = v;
        if (xedge) xedge[(((int64_t)b * G + g) * Cig + ci) * 2 + k] = v;      // kept for the weight gradient's row-end terms
    }
    __syncthreads();
    // a wave takes FOUR channels at a time (lanes over the input channels, coalesced weight rows): their 16 loads are in flight
    // together, then the eight wave sums -- one channel at a time made the kernel a chain of dependent latencies (48 us)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c4 = 4 * wave; c4 < Cr; c4 += 16) {
        float c0[4] = {0.f, 0.f, 0.f, 0.f}, cl[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ci = lane; ci < Cig; ci += 64) {
            const float x0 = xl[ci], xe = xl[Cig + ci];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (c4 + u < Cr) {
                    const float* const wr = w + (((int64_t)g * Cr + c4 + u) * Cig + ci) * 3;
                    c0[u] = fmaf(wr[0], x0, c0[u]);
                    cl[u] = fmaf(wr[2], xe, cl[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) c0[u] = nef_wave_sum(c0[u]), cl[u] = nef_wave_sum(cl[u]);
        if (lane < 4 && c4 + lane < Cr) {
            const float c0v = lane == 0 ? c0[0] : (lane == 1 ? c0[1] : (lane == 2 ? c0[2] : c0[3]));
            const float clv = lane == 0 ? cl[0] : (lane == 1 ? cl[1] : (lane == 2 ? cl[2] : cl[3]));
            const int64_t ch = (int64_t)g * Cr + c4 + lane;
            float* const row = y + ((int64_t)b * G * Cr + ch) * T;
            const float y0 = row[0], yl = row[T - 1];
            row[0] = y0 - c0v;
            row[T - 1] = yl - clv;
            if (slots) {
                float* const sp = slots + ((ch * B + b) * nslot) * 2;
                sp[0] -= c0v + clv;
                sp[1] += fmaf(c0v, c0v - 2.f * y0, clv * (clv - 2.f * yl));
            }
        }
    }
}

// Weight gradient of the polyphase form: gw2 [R 2][Cig][3] = the gradients of the phase weights (row 2 r + p), folded back through
// W'_p(w) -- the transpose of poly_weights_kernel's 6 x 3 map -- minus the two row-end terms of the forward correction:
//     gw[r][ci][0] -= sum_b gy[b][r][0] x'[b][ci][0]        gw[r][ci][2] -= sum_b gy[b][r][T-1] x'[b][ci][Tin-1]
// gy phase-major [B][G 2 Cog][Tin], xedge [B][G Cig][2] (nef_poly_fwd_edge).  Two launches: the row-end sums per (row r = g Cog + co,
// chunk of POLY_BC samples) -> part [chunks][R][Cig][2] (fixed order, no atomics), then the fold, one workgroup per row.
constexpr int POLY_BC = 64;
__global__ __launch_bounds__(256) void poly_wgrad_edge_partial(const float* __restrict__ gy, const float* __restrict__ xedge,
                                                               float* __restrict__ part, int B, int G, int Cog, int Cig, int Tin) {
    __shared__ float ge[2][POLY_BC];
    const int r = blockIdx.x, g = r / Cog, co = r - g * Cog;
    const int b0 = blockIdx.y * POLY_BC;
    const int nb = B - b0 < POLY_BC ? B - b0 : POLY_BC;
    if (threadIdx.x < 2 * POLY_BC) {
        const int i = threadIdx.x >> 1, p = threadIdx.x & 1;
        ge[p][i] = i < nb ? gy[(((int64_t)(b0 + i) * G + g) * 2 * Cog + 2 * co + p) * Tin + (p ? Tin - 1 : 0)] : 0.f;
    }
    __syncthreads();
    for (int ci = threadIdx.x; ci < Cig; ci += 256) {
        const float* xe = xedge + (((int64_t)b0 * G + g) * Cig + ci) * 2;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll 8
        for (int b = 0; b < nb; ++b) {
            const nef_f32x2 xv = *(const nef_f32x2*)(xe + (int64_t)b * G * Cig * 2);
            s0 = fmaf(ge[0][b], xv[0], s0);
            s1 = fmaf(ge[1][b], xv[1], s1);
        }
        *(nef_f32x2*)(part + (((int64_t)blockIdx.y * gridDim.x + r) * Cig + ci) * 2) = nef_f32x2{s0, s1};
    }
}

__global__ __launch_bounds__(256) void poly_wgrad_fold_kernel(const float* __restrict__ gw2, const float* __restrict__ part,
                                                              float* __restrict__ gw, int R, int Cig, int chunks) {
    const int r = blockIdx.x;
    for (int ci = threadIdx.x; ci < Cig; ci += 256) {
        float e0 = 0.f, el = 0.f;
        for (int c = 0; c < chunks; ++c) {
            const nef_f32x2 v = *(const nef_f32x2*)(part + (((int64_t)c * R + r) * Cig + ci) * 2);
            e0 += v[0], el += v[1];
        }
        const float* A = gw2 + ((int64_t)(2 * r) * Cig + ci) * 3;
        const float* Bq = gw2 + ((int64_t)(2 * r + 1) * Cig + ci) * 3;
        const float a0 = A[0], a1 = A[1], a2 = A[2], b0_ = Bq[0], b1 = Bq[1], b2 = Bq[2];
        float* o = gw + ((int64_t)r * Cig + ci) * 3;
        o[0] = fmaf(0.75f, a0 + b1, 0.25f * (a1 + b0_)) - e0;
        o[1] = fmaf(0.75f, a1 + b1, 0.25f * (a0 + b2));
        o[2] = fmaf(0.75f, a1 + b2, 0.25f * (a2 + b1)) - el;
    }
}

// Row ends of the polyphase backward-data pass.  The phase convs treat both ends of x as if the interpolation formula continued
// (x[-1] = x[0], x[Tin] = x[Tin-1]: nn.Upsample's clamped sources) and as if u[-1], u[T] existed; the conv's zero padding of u
// says they do not.  Written out (gu = the gradient wrt u the full-resolution pass would have produced):
//     gx[0]      += 0.25 (w1 - w0)^T gy[:, 0]   + 0.25 w0^T gy[:, 1]
//     gx[Tin-1]  += 0.25 (w1 - w2)^T gy[:, T-1] + 0.25 w2^T gy[:, T-2]
// One workgroup per (sample, group); also adds the two columns' share of the BatchNorm-backward sums to the sample's first slot.
__global__ __launch_bounds__(256) void poly_bwd_edge_kernel(const float* __restrict__ gy, const float* __restrict__ w, float* __restrict__ gx,
                                                            int B, int G, int Cog, int Cig, int T, const float* __restrict__ bx,
                                                            const float* __restrict__ bmean, const float* __restrict__ binv,
                                                            const float* __restrict__ ba, const float* __restrict__ bb, int Bp,
                                                            float* __restrict__ slots, int nslot, int pm) {
    extern __shared__ float gl[];      // [4][Cog]: columns 0, 1, T-2, T-1 of this (sample, group)'s gradient rows
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const int Tin = T >> 1;
    const float* const gyb = gy + ((int64_t)b * G + g) * Cog * T;
    for (int i = threadIdx.x; i < 4 * Cog; i += 256) {
        const int c = i >> 2, k = i & 3;
        const int t = k < 2 ? k : T - 4 + k;
        // pm: gy is stored phase-major, [2 Cog][T / 2] -- column t of channel c is element t >> 1 of row 2 c + (t & 1)
        gl[k * Cog + c] = pm ? gyb[(int64_t)(2 * c + (t & 1)) * Tin + (t >> 1)] : gyb[(int64_t)c * T + t];
    }
    __syncthreads();
    const int64_t ctot = (int64_t)G * Cig;
    for (int ci = threadIdx.x; ci < Cig; ci += 256) {
        float d0 = 0.f, dl = 0.f;
        const float* wp = w + ((int64_t)g * Cog * Cig + ci) * 3;
        for (int co = 0; co < Cog; ++co, wp += (int64_t)Cig * 3) {
            const float w0 = wp[0], w1 = wp[1], w2 = wp[2];
            d0 = fmaf(w1 - w0, gl[co], fmaf(w0, gl[Cog + co], d0));
            dl = fmaf(w1 - w2, gl[3 * Cog + co], fmaf(w2, gl[2 * Cog + co], dl));
        }
        d0 *= 0.25f, dl *= 0.25f;
        const int64_t ch = (int64_t)g * Cig + ci;
        float* const row = gx + ((int64_t)b * ctot + ch) * Tin;
        row[0] += d0;
        row[Tin - 1] += dl;
        if (slots) {
            const int64_t pr = (int64_t)(b / Bp) * ctot + ch;
            const float* const xr = bx + ((int64_t)b * ctot + ch) * Tin;
            const float af = ba[pr], bf = bb[pr], mf = bmean[pr], is = binv[pr];
            const float x0 = xr[0], xl = xr[Tin - 1];
            const float g0 = fmaf(x0, af, bf) > 0.f ? d0 : 0.f, g1 = fmaf(xl, af, bf) > 0.f ? dl : 0.f;
            float* const sl = slots + ((ch * B + b) * nslot) * 2;
            sl[0] += g0 + g1;
            sl[1] += fmaf(g0, (x0 - mf) * is, g1 * ((xl - mf) * is));
        }
    }
}

// out[b][c] = sum_s slots[c][b * nslot + s][0]: the per-(sample, channel) sums a conv epilogue left in stats_mode 1
__global__ __launch_bounds__(256) void slots_to_rows_kernel(const float* __restrict__ slots, int nslot, float* __restrict__ out, int B, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * C) return;
    const int b = (int)(i / C), c = (int)(i - (int64_t)b * C);
    const float* p = slots + (((int64_t)c * B + b) * nslot) * 2;
    float s = 0.f;
    for (int k = 0; k < nslot; ++k) s += p[2 * k];      // fixed order
    out[i] = s;
}

__global__ void h2_taint_kernel(const int32_t* __restrict__ total, int32_t* __restrict__ mark, float* __restrict__ out) {
    const int32_t t = total[0];
    out[0] = (float)(t - mark[0]);
    mark[0] = t;
}

}  // namespace

#define NEF_ST ((hipStream_t)stream)

extern "C" {

int nef_mix_bwd_shared(const float* gD2, const float* latent, const float* z1, const float* z2r, const float* q,
                       float* gz1, float* gz2r, float* gq, int B, int V, int T, int c1, int c2,
                       const int32_t* choice_dev, int relu_z1, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(gD2 && latent && z1 && z2r && q && gz1 && gz2r && gq, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && V > 0 && T > 1 && c1 >= 0 && c1 < V && c2 >= 0 && c2 < V, NEF_E_SHAPE);
    const bool al16 = (((uintptr_t)gD2 | (uintptr_t)latent | (uintptr_t)z1 | (uintptr_t)z2r | (uintptr_t)gz1 | (uintptr_t)gz2r) & 15) == 0;
    if (T % 4 == 2 && T >= 8 && al16)
        hipLaunchKernelGGL(mix_bwd_shared_pair_kernel, dim3(nef_stream_grid((int64_t)B * 128, 4)), dim3(256), 0, NEF_ST, gD2, latent, z1,
                           z2r, q, gz1, gz2r, gq, B, V, T, c1, c2, choice_dev, relu_z1);
    else
        hipLaunchKernelGGL((mix_bwd_kernel<false, true>), dim3(nef_stream_grid((int64_t)B * 256, 4)), dim3(256), 0, NEF_ST, gD2,
                           latent, z1, z2r, q, gz1, gz2r, gq, B, V, T, c1, c2, choice_dev, relu_z1);
    return nef_launch_status();
}

int nef_mix_bwd_shared_up(const float* gU2, const float* latent, const float* z1, const float* z2r, const float* q,
                          float* gz1, float* gz2r, float* gq, int B, int V, int T, int c1, int c2,
                          const int32_t* choice_dev, int relu_z1, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(gU2 && latent && z1 && z2r && q && gz1 && gz2r && gq, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && V > 0 && T > 1 && c1 >= 0 && c1 < V && c2 >= 0 && c2 < V, NEF_E_SHAPE);
    hipLaunchKernelGGL((mix_bwd_kernel<true, true>), dim3(nef_stream_grid((int64_t)B * 256, 4)), dim3(256), 0, NEF_ST, gU2,
                       latent, z1, z2r, q, gz1, gz2r, gq, B, V, T, c1, c2, choice_dev, relu_z1);
    return nef_launch_status();
}

// ------------------------------------------------------------------------------------------------
// First decoder conv shared between the three Standin passes: P2 [2B][2C][L] holds, per sample n, the half-conv of
// the z1-type half (channels 0..C) and of the z2-type half (C..2C) of input n (n < B: the lead means, n >= B: the picked
// leads).  c1[p] = A-half[ia(p)] + B-half[ib(p)] + bias with (ia, ib) = (mean, mean), (pick, mean), (mean, pick).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pass_combine_fwd_kernel(const float* __restrict__ P2, const float* __restrict__ bias,
                                                               float* __restrict__ c1, int B, int C, int L) {
    const int64_t row = blockIdx.x;                 // (b, c)
    const int b = (int)(row / C), c = (int)(row % C);
    const float bv = bias[c];
    const float* am = P2 + ((int64_t)b * 2 * C + c) * L;
    const float* bm = P2 + ((int64_t)b * 2 * C + C + c) * L;
    const float* ap = P2 + ((int64_t)(B + b) * 2 * C + c) * L;
    const float* bp = P2 + ((int64_t)(B + b) * 2 * C + C + c) * L;
    float* o0 = c1 + ((int64_t)b * C + c) * L;
    float* o1 = c1 + ((int64_t)(B + b) * C + c) * L;
    float* o2 = c1 + ((int64_t)(2 * B + b) * C + c) * L;
    for (int t = threadIdx.x; t < L; t += 256) {
        const float a0 = am[t], b0 = bm[t], a1 = ap[t], b1 = bp[t];
        o0[t] = a0 + b0 + bv;
        o1[t] = a1 + b0 + bv;
        o2[t] = a0 + b1 + bv;
    }
}

// The same pass, also leaving the BatchNorm statistics of its output (decoder.1.double_conv.1, batch statistics per
// Standin pass): per (sample, channel) row the three passes' sum and sum of squares in fp64,
// part[((p*C + c)*B + b)*2 + {0,1}]; bn_stats_final adds them over b in a fixed order.  Saves the separate
// statistics pass over c1.
__global__ __launch_bounds__(256) void pass_combine_fwd_stats_kernel(const float* __restrict__ P2,
                                                                     const float* __restrict__ bias,
                                                                     float* __restrict__ c1, double* __restrict__ part,
                                                                     int B, int C, int L) {
    __shared__ double sm[4];
    const int64_t row = blockIdx.x;                 // (b, c)
    const int b = (int)(row / C), c = (int)(row % C);
    const float bv = bias[c];
    const float* am = P2 + ((int64_t)b * 2 * C + c) * L;
    const float* bm = P2 + ((int64_t)b * 2 * C + C + c) * L;
    const float* ap = P2 + ((int64_t)(B + b) * 2 * C + c) * L;
    const float* bp = P2 + ((int64_t)(B + b) * 2 * C + C + c) * L;
    float* o0 = c1 + ((int64_t)b * C + c) * L;
    float* o1 = c1 + ((int64_t)(B + b) * C + c) * L;
    float* o2 = c1 + ((int64_t)(2 * B + b) * C + c) * L;
    double s[3][2] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};
    for (int t = threadIdx.x; t < L; t += 256) {
        const float a0 = am[t], b0 = bm[t], a1 = ap[t], b1 = bp[t];
        const float v0 = a0 + b0 + bv, v1 = a1 + b0 + bv, v2 = a0 + b1 + bv;
        o0[t] = v0;
        o1[t] = v1;
        o2[t] = v2;
        s[0][0] += (double)v0; s[0][1] += (double)v0 * (double)v0;
        s[1][0] += (double)v1; s[1][1] += (double)v1 * (double)v1;
        s[2][0] += (double)v2; s[2][1] += (double)v2 * (double)v2;
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const double r = nef_block_sum_d(s[p][q], sm);
            if (threadIdx.x == 0) part[(((int64_t)p * C + c) * B + b) * 2 + q] = r;
        }
}

// tot[pc][q] = sum_b rows[pc][b][q], one block per (pass, channel), fixed tree order (deterministic)
__global__ __launch_bounds__(256) void bn_rows_reduce(const double* __restrict__ rows, double* __restrict__ tot, int B) {
    __shared__ double sm[4];
    const int64_t pc = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int b = threadIdx.x; b < B; b += 256) {
        s1 += rows[(pc * B + b) * 2];
        s2 += rows[(pc * B + b) * 2 + 1];
    }
    s1 = nef_block_sum_d(s1, sm);
    s2 = nef_block_sum_d(s2, sm);
    if (threadIdx.x == 0) {
        tot[pc * 2] = s1;
        tot[pc * 2 + 1] = s2;
    }
}

// tot[(p*C + c)*2 + q] = sum over the slots of pass p of slots[c][p*per_pass + i][q] (the conv epilogue's fp32 slot sums,
// nef_conv_args.stats), in fp64 and in a fixed order
__global__ __launch_bounds__(256) void bn_slots_reduce(const float* __restrict__ slots, double* __restrict__ tot, int P,
                                                       int C, int per_pass) {
    __shared__ double sm[4];
    const int c = blockIdx.x % C, p = blockIdx.x / C;
    const float2* src = (const float2*)slots + ((int64_t)c * P + p) * per_pass;
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < per_pass; i += 256) {
        const float2 v = src[i];
        s1 += (double)v.x;
        s2 += (double)v.y;
    }
    s1 = nef_block_sum_d(s1, sm);
    s2 = nef_block_sum_d(s2, sm);
    if (threadIdx.x == 0) {
        tot[((int64_t)p * C + c) * 2] = s1;
        tot[((int64_t)p * C + c) * 2 + 1] = s2;
    }
}

// Round 6: bn_slots_reduce + bn_stats_final (forward) / + bn_bwd_final (backward) as ONE launch -- one workgroup per CHANNEL walks
// its P passes: each pass's slot sums are block-reduced in fp64 in a fixed order (1024 threads, eight loads in flight each) and thread 0
// then does that pass's share of the final (running statistics pass by pass, in order).  Two launches fewer per BatchNorm and direction.
constexpr int BNF_THREADS = 1024;      // sixteen waves per channel: the slots of a pass are two round trips of eight loads per thread
__device__ __forceinline__ void bn_slot_totals(const float* __restrict__ slots, int c, int p, int P, int per_pass, double* sm,
                                               double& s1, double& s2) {
    const float2* src = (const float2*)slots + ((int64_t)c * P + p) * per_pass;
    s1 = 0.0, s2 = 0.0;
    // eight loads in flight per thread, added in index order.  One workgroup per channel is all the parallelism this pass has
    // (64-128 workgroups): with 256 threads and a load per trip it took 37 us for 15 MB -- ~2 us of latency per dependent trip
    for (int i0 = threadIdx.x; i0 < per_pass; i0 += 8 * BNF_THREADS) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * BNF_THREADS;
            v[u] = i < per_pass ? src[i] : float2{0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s1 += (double)v[u].x;
            s2 += (double)v[u].y;
        }
    }
    // block sum over sixteen waves, fixed order (deterministic): wave sums, then wave 0's order 0..15
    s1 = nef_wave_sum_d(s1);
    s2 = nef_wave_sum_d(s2);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[2 * w] = s1, sm[2 * w + 1] = s2;
    __syncthreads();
    s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < BNF_THREADS / 64; ++k) s1 += sm[2 * k], s2 += sm[2 * k + 1];
}

__global__ __launch_bounds__(BNF_THREADS) void bn_slots_stats_fused(const float* __restrict__ slots, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ running_mean,
                                                            float* __restrict__ running_var, float* __restrict__ mean,
                                                            float* __restrict__ invstd, float* __restrict__ a, float* __restrict__ b, int P,
                                                            int Bp, int C, int L, float eps, float momentum, int per_pass,
                                                            int64_t* __restrict__ nbt) {
    __shared__ double sm[2 * BNF_THREADS / 64];
    const int c = blockIdx.x;
    const double n = (double)Bp * (double)L;
    float rm = running_mean ? running_mean[c] : 0.f;
    float rv = running_var ? running_var[c] : 1.f;
    for (int p = 0; p < P; ++p) {
        double s1, s2;
        bn_slot_totals(slots, c, p, P, per_pass, sm, s1, s2);
        if (threadIdx.x == 0) {      // (bn_stats_final's arithmetic)
            const double m = s1 / n;
            double var = s2 / n - m * m;
            if (var < 0.0) var = 0.0;
            const float mf = (float)m;
            const float is = (float)(1.0 / sqrt(var + (double)eps));
            mean[p * C + c] = mf;
            invstd[p * C + c] = is;
            const float af = gamma[c] * is;
            a[p * C + c] = af;
            b[p * C + c] = beta[c] - mf * af;
            const float unbiased = (float)(var * (n / (n - 1.0)));
            rm = (1.f - momentum) * rm + momentum * mf;
            rv = (1.f - momentum) * rv + momentum * unbiased;
        }
    }
    if (threadIdx.x == 0) {
        if (running_mean) running_mean[c] = rm;
        if (running_var) running_var[c] = rv;
        if (nbt && c == 0) nbt[0] += P;
    }
}

__global__ __launch_bounds__(BNF_THREADS) void bn_slots_bwd_fused(const float* __restrict__ slots, float* __restrict__ coef,
                                                          float* __restrict__ ggamma, float* __restrict__ gbeta, int P, int Bp, int C,
                                                          int L, int per_pass) {
    __shared__ double sm[2 * BNF_THREADS / 64];
    const int c = blockIdx.x;
    const double n = (double)Bp * (double)L;
    double g1 = 0.0, g2 = 0.0;
    for (int p = 0; p < P; ++p) {
        double s1, s2;
        bn_slot_totals(slots, c, p, P, per_pass, sm, s1, s2);
        if (threadIdx.x == 0) {      // (bn_bwd_final's arithmetic)
            coef[(p * C + c) * 2] = (float)(s1 / n);
            coef[(p * C + c) * 2 + 1] = (float)(s2 / n);
            g1 += s1;
            g2 += s2;
        }
    }
    if (threadIdx.x == 0) {
        gbeta[c] = (float)g1;
        ggamma[c] = (float)g2;
    }
}

// adjoint: gP2 A-half[mean] = g0 + g2, A-half[pick] = g1, B-half[mean] = g0 + g1, B-half[pick] = g2
__global__ __launch_bounds__(256) void pass_combine_bwd_kernel(const float* __restrict__ gc1, float* __restrict__ gP2,
                                                               int B, int C, int L) {
    const int64_t row = blockIdx.x;
    const int b = (int)(row / C), c = (int)(row % C);
    const float* g0 = gc1 + ((int64_t)b * C + c) * L;
    const float* g1 = gc1 + ((int64_t)(B + b) * C + c) * L;
    const float* g2 = gc1 + ((int64_t)(2 * B + b) * C + c) * L;
    float* am = gP2 + ((int64_t)b * 2 * C + c) * L;
    float* bm = gP2 + ((int64_t)b * 2 * C + C + c) * L;
    float* ap = gP2 + ((int64_t)(B + b) * 2 * C + c) * L;
    float* bp = gP2 + ((int64_t)(B + b) * 2 * C + C + c) * L;
    for (int t = threadIdx.x; t < L; t += 256) {
        const float x0 = g0[t], x1 = g1[t], x2 = g2[t];
        am[t] = x0 + x2;
        ap[t] = x1;
        bm[t] = x0 + x1;
        bp[t] = x2;
    }
}

int nef_pass_combine_fwd(const float* P2, const float* bias, float* c1, int B, int C, int L, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(P2 && bias && c1, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && C > 0 && L > 0 && (int64_t)B * C <= 0x7FFFFFFF, NEF_E_SHAPE);
    hipLaunchKernelGGL(pass_combine_fwd_kernel, dim3((unsigned)(B * C)), dim3(256), 0, NEF_ST, P2, bias, c1, B, C, L);
    return nef_launch_status();
}

int nef_pass_combine_bwd(const float* gc1, float* gP2, int B, int C, int L, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(gc1 && gP2, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && C > 0 && L > 0 && (int64_t)B * C <= 0x7FFFFFFF, NEF_E_SHAPE);
    hipLaunchKernelGGL(pass_combine_bwd_kernel, dim3((unsigned)(B * C)), dim3(256), 0, NEF_ST, gc1, gP2, B, C, L);
    return nef_launch_status();
}

int nef_chscale_fwd(const float* x, const float* s, int64_t s_bs, float* y, int B, int C, int T, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(x && s && y, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && C > 0 && T > 0, NEF_E_SHAPE);
    hipLaunchKernelGGL(chscale_fwd_kernel, dim3(nef_stream_grid((int64_t)B * C, 4)), dim3(256), 0, NEF_ST, x, s, s_bs,
                       y, B, C, T);
    return nef_launch_status();
}

int nef_chscale_bwd(const float* gy, const float* x, const float* s, int64_t s_bs, float* gx, float* gs, int B, int C,
                    int T, int relu_x, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(gy && x && s && gx && gs, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && C > 0 && T > 0, NEF_E_SHAPE);
    hipLaunchKernelGGL(chscale_bwd_kernel, dim3(nef_stream_grid((int64_t)B * C, 4)), dim3(256), 0, NEF_ST, gy, x, s,
                       s_bs, gx, gs, B, C, T, relu_x);
    return nef_launch_status();
}

int nef_gate(const float* g, const float* ref, float* out, float scale, int64_t n, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(g && ref && out, NEF_E_NULL);
    NEF_REQUIRE(n > 0, NEF_E_SHAPE);
    if ((n & 3) == 0 && (((uintptr_t)g | (uintptr_t)ref | (uintptr_t)out) & 15) == 0) {
        hipLaunchKernelGGL(gate4_kernel, dim3(nef_stream_grid(n >> 2, 256)), dim3(256), 0, NEF_ST, (const nef_f32x4*)g,
                           (const nef_f32x4*)ref, (nef_f32x4*)out, scale, n >> 2);
        return nef_launch_status();
    }
    hipLaunchKernelGGL(gate_kernel, dim3(nef_stream_grid(n, 256)), dim3(256), 0, NEF_ST, g, ref, out, scale, n);
    return nef_launch_status();
}

int nef_add(const float* a, const float* b, float* out, int64_t n, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(a && b && out, NEF_E_NULL);
    NEF_REQUIRE(n > 0, NEF_E_SHAPE);
    hipLaunchKernelGGL(add_kernel, dim3(nef_stream_grid(n, 256)), dim3(256), 0, NEF_ST, a, b, out, n);
    return nef_launch_status();
}

int nef_window_crop(const float* src, int64_t x_bs, int64_t x_gs, float* dst, int B, int G, int Cg, int T, int t0, int W,
                    nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(src && dst, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && G > 0 && Cg > 0 && W > 0 && t0 >= 0 && t0 + W <= T, NEF_E_SHAPE);
    const int64_t n = (int64_t)B * G * Cg * W;
    hipLaunchKernelGGL(window_crop_kernel, dim3(nef_stream_grid(n, 256)), dim3(256), 0, NEF_ST, src, x_bs, x_gs, dst, B,
                       G, Cg, T, t0, W);
    return nef_launch_status();
}

int nef_window_scatter(const float* src, float* dst, int64_t y_bs, int64_t y_gs, int B, int G, int Cg, int T, int t0,
                       int W, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(src && dst, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && G > 0 && Cg > 0 && W > 0 && t0 >= 0 && t0 + W <= T, NEF_E_SHAPE);
    hipLaunchKernelGGL(window_scatter_kernel, dim3(nef_stream_grid((int64_t)B * G * Cg, 4)), dim3(256), 0, NEF_ST, src,
                       dst, y_bs, y_gs, B, G, Cg, T, t0, W);
    return nef_launch_status();
}

int nef_lead_mean(const float* z1, const float* z2r, float* latent, int B, int V, int T, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(z1 && z2r && latent, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && V > 0 && T > 0, NEF_E_SHAPE);
    hipLaunchKernelGGL(lead_mean_kernel, dim3(nef_stream_grid((int64_t)B * 256, 4)), dim3(256), 0, NEF_ST, z1, z2r,
                       latent, B, V, T);
    return nef_launch_status();
}

int nef_mix_fwd(const float* latent, const float* z1, const float* z2r, const float* q, float* D, int B, int V, int T,
                int c1, int c2, const int32_t* choice_dev, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(latent && z1 && z2r && q && D, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && V > 0 && T > 0 && c1 >= 0 && c1 < V && c2 >= 0 && c2 < V, NEF_E_SHAPE);
    hipLaunchKernelGGL(mix_fwd_kernel<false>, dim3(nef_stream_grid((int64_t)B * 256, 4)), dim3(256), 0, NEF_ST, latent, z1,
                       z2r, q, D, B, V, T, c1, c2, choice_dev);
    return nef_launch_status();
}

int nef_mix_fwd_shared(const float* latent, const float* z1, const float* z2r, const float* q, float* D2, int B, int V,
                       int T, int c1, int c2, const int32_t* choice_dev, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(latent && z1 && z2r && q && D2, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && V > 0 && T > 0 && c1 >= 0 && c1 < V && c2 >= 0 && c2 < V, NEF_E_SHAPE);
    hipLaunchKernelGGL(mix_fwd_kernel<true>, dim3(nef_stream_grid((int64_t)B * 256, 4)), dim3(256), 0, NEF_ST, latent, z1,
                       z2r, q, D2, B, V, T, c1, c2, choice_dev);
    return nef_launch_status();
}

int nef_lead_mean_mix_shared(const float* z1, const float* z2r, const float* q, float* latent, float* D2, int B, int V,
                             int T, int c1, int c2, const int32_t* choice_dev, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(z1 && z2r && q && latent && D2, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && V > 0 && T > 0 && c1 >= 0 && c1 < V && c2 >= 0 && c2 < V, NEF_E_SHAPE);
    const dim3 grid(nef_stream_grid((int64_t)B * 256, 4));
    const bool al16 = (((uintptr_t)z1 | (uintptr_t)z2r | (uintptr_t)latent | (uintptr_t)D2) & 15) == 0;
    if (T % 4 == 2 && al16)
        hipLaunchKernelGGL(lead_mean_mix_shared_pair_kernel, dim3(nef_stream_grid((int64_t)B * 128, 4)), dim3(256), 0, NEF_ST, z1, z2r, q,
                           latent, D2, B, V, T, c1, c2, choice_dev);
    else if (T % 2 == 0)
        hipLaunchKernelGGL(lead_mean_mix_shared_kernel<2>, grid, dim3(256), 0, NEF_ST, z1, z2r, q, latent, D2, B, V, T, c1,
                           c2, choice_dev);
    else
        hipLaunchKernelGGL(lead_mean_mix_shared_kernel<1>, grid, dim3(256), 0, NEF_ST, z1, z2r, q, latent, D2, B, V, T, c1,
                           c2, choice_dev);
    return nef_launch_status();
}

int nef_mix_bwd(const float* gD, const float* latent, const float* z1, const float* z2r, const float* q, float* gz1,
                float* gz2r, float* gq, int B, int V, int T, int c1, int c2, const int32_t* choice_dev, int relu_z1,
                nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(gD && latent && z1 && z2r && q && gz1 && gz2r && gq, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && V > 0 && T > 0 && c1 >= 0 && c1 < V && c2 >= 0 && c2 < V, NEF_E_SHAPE);
    hipLaunchKernelGGL((mix_bwd_kernel<false, false>), dim3(nef_stream_grid((int64_t)B * 256, 4)), dim3(256), 0, NEF_ST, gD, latent,
                       z1, z2r, q, gz1, gz2r, gq, B, V, T, c1, c2, choice_dev, relu_z1);
    return nef_launch_status();
}

int nef_mix_bwd_up(const float* gU, const float* latent, const float* z1, const float* z2r, const float* q, float* gz1,
                   float* gz2r, float* gq, int B, int V, int T, int c1, int c2, const int32_t* choice_dev, int relu_z1,
                   nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(gU && latent && z1 && z2r && q && gz1 && gz2r && gq, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && V > 0 && T > 1 && c1 >= 0 && c1 < V && c2 >= 0 && c2 < V, NEF_E_SHAPE);
    hipLaunchKernelGGL((mix_bwd_kernel<true, false>), dim3(nef_stream_grid((int64_t)B * 256, 4)), dim3(256), 0, NEF_ST, gU, latent,
                       z1, z2r, q, gz1, gz2r, gq, B, V, T, c1, c2, choice_dev, relu_z1);
    return nef_launch_status();
}

int nef_upsample2_fwd(const float* x, float* y, int64_t N, int Tin, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(x && y, NEF_E_NULL);
    NEF_REQUIRE(N > 0 && Tin > 0, NEF_E_SHAPE);
    hipLaunchKernelGGL(upsample2_fwd_kernel, dim3(nef_stream_grid(N, 4)), dim3(256), 0, NEF_ST, x, y, N, Tin);
    return nef_launch_status();
}

int nef_upsample2_aff_fwd(const float* x, const float* a, const float* b, float* y, int N, int C, int Tin, int Bp,
                          nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(x && a && b && y, NEF_E_NULL);
    NEF_REQUIRE(N > 0 && C > 0 && Tin > 0 && Bp > 0, NEF_E_SHAPE);
    const int64_t rows = (int64_t)N * C;
    if ((Tin & 1) == 0 && Tin >= 4 && rows <= 0x7FFFFFFF) {
        hipLaunchKernelGGL(upsample2_aff_fwd_rows, dim3((unsigned)rows), dim3(256), 0, NEF_ST, x, a, b, y, C, Tin, Bp);
        return nef_launch_status();
    }
    hipLaunchKernelGGL(upsample2_aff_fwd_kernel, dim3(nef_stream_grid(rows, 4)), dim3(256), 0, NEF_ST, x, a, b, y, rows, C,
                       Tin, Bp);
    return nef_launch_status();
}

int nef_upsample2_bwd(const float* gy, float* gx, int64_t N, int Tin, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(gy && gx, NEF_E_NULL);
    NEF_REQUIRE(N > 0 && Tin > 0, NEF_E_SHAPE);
    if ((Tin & 1) == 0 && Tin >= 4 && N <= 0x7FFFFFFF) {
        hipLaunchKernelGGL(upsample2_bwd_rows, dim3((unsigned)N), dim3(256), 0, NEF_ST, gy, gx, Tin);
        return nef_launch_status();
    }
    hipLaunchKernelGGL(upsample2_bwd_kernel, dim3(nef_stream_grid(N, 4)), dim3(256), 0, NEF_ST, gy, gx, N, Tin);
    return nef_launch_status();
}

size_t nef_bn_ws_bytes(int P, int C) {
    return (size_t)P * C * BN_SPLIT * 2 * sizeof(double) + (size_t)P * C * 2 * sizeof(float);
}

int nef_bn_train_stats(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                       float* mean, float* invstd, float* a, float* b, void* ws, size_t ws_bytes, int P, int Bp, int C,
                       int L, float eps, float momentum, int64_t* num_batches_tracked, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(x && gamma && beta && mean && invstd && a && b && ws, NEF_E_NULL);
    NEF_REQUIRE(P > 0 && Bp > 0 && C > 0 && L > 0 && (int64_t)Bp * L > 1, NEF_E_SHAPE);
    NEF_REQUIRE(ws_bytes >= nef_bn_ws_bytes(P, C), NEF_E_WORKSPACE);
    hipLaunchKernelGGL(bn_stats_partial, dim3(P * C * BN_SPLIT), dim3(256), 0, NEF_ST, x, (double*)ws, P, Bp, C, L);
    hipLaunchKernelGGL(bn_stats_final, dim3((C + 63) / 64), dim3(64), 0, NEF_ST, (const double*)ws, gamma, beta,
                       running_mean, running_var, mean, invstd, a, b, P, Bp, C, L, eps, momentum, BN_SPLIT, num_batches_tracked);
    return nef_launch_status();
}

int nef_bn_stats_from_slots(const float* slots, int nslot, const float* gamma, const float* beta, float* running_mean,
                            float* running_var, float* mean, float* invstd, float* a, float* b, void* ws,
                            size_t ws_bytes, int P, int Bp, int C, int L, float eps, float momentum,
                            int64_t* num_batches_tracked, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(slots && gamma && beta && mean && invstd && a && b && ws, NEF_E_NULL);
    NEF_REQUIRE(P > 0 && Bp > 0 && C > 0 && L > 0 && nslot > 0 && (int64_t)Bp * L > 1 &&
                    (int64_t)Bp * nslot <= 0x7FFFFFFF && (int64_t)P * C <= 0x7FFFFFFF, NEF_E_SHAPE);
    NEF_REQUIRE(ws_bytes >= nef_bn_ws_bytes(P, C), NEF_E_WORKSPACE);
    hipLaunchKernelGGL(bn_slots_stats_fused, dim3((unsigned)C), dim3(BNF_THREADS), 0, NEF_ST, slots, gamma, beta, running_mean, running_var, mean,
                       invstd, a, b, P, Bp, C, L, eps, momentum, Bp * nslot, num_batches_tracked);
    return nef_launch_status();
}

size_t nef_pass_combine_stats_ws_bytes(int B, int C) { return (size_t)3 * C * (B + 1) * 2 * sizeof(double); }

int nef_pass_combine_fwd_stats(const float* P2, const float* bias, float* c1, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float* mean, float* invstd, float* a, float* b,
                               void* ws, size_t ws_bytes, int B, int C, int L, float eps, float momentum,
                               int64_t* num_batches_tracked, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(P2 && bias && c1 && gamma && beta && mean && invstd && a && b && ws, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && C > 0 && L > 0 && (int64_t)B * C <= 0x7FFFFFFF && (int64_t)B * L > 1, NEF_E_SHAPE);
    NEF_REQUIRE(ws_bytes >= nef_pass_combine_stats_ws_bytes(B, C), NEF_E_WORKSPACE);
    double* rows = (double*)ws;                      // [3*C][B][2]
    double* tot = rows + (size_t)3 * C * B * 2;      // [3*C][2]
    hipLaunchKernelGGL(pass_combine_fwd_stats_kernel, dim3((unsigned)(B * C)), dim3(256), 0, NEF_ST, P2, bias, c1, rows, B,
                       C, L);
    hipLaunchKernelGGL(bn_rows_reduce, dim3((unsigned)(3 * C)), dim3(256), 0, NEF_ST, (const double*)rows, tot, B);
    hipLaunchKernelGGL(bn_stats_final, dim3((C + 63) / 64), dim3(64), 0, NEF_ST, (const double*)tot, gamma, beta,
                       running_mean, running_var, mean, invstd, a, b, 3, B, C, L, eps, momentum, 1, num_batches_tracked);
    return nef_launch_status();
}

int nef_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float* a, float* b, int C, float eps, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(gamma && beta && running_mean && running_var && a && b, NEF_E_NULL);
    NEF_REQUIRE(C > 0, NEF_E_SHAPE);
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((C + 63) / 64), dim3(64), 0, NEF_ST, gamma, beta, running_mean,
                       running_var, a, b, C, eps);
    return nef_launch_status();
}

int nef_fold_bn(const float* w, const float* bias, const float* a, const float* b, float* w_out, float* bias_out,
                int Cout, int inner, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(w && bias && a && b && w_out && bias_out, NEF_E_NULL);
    NEF_REQUIRE(Cout > 0 && inner > 0, NEF_E_SHAPE);
    hipLaunchKernelGGL(fold_bn_kernel, dim3(nef_stream_grid((int64_t)Cout * inner, 256)), dim3(256), 0, NEF_ST, w, bias,
                       a, b, w_out, bias_out, Cout, inner);
    return nef_launch_status();
}

int nef_affine_relu_fwd(const float* x, const float* a, const float* b, float* y, int P, int Bp, int C, int L,
                        nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(x && a && b && y, NEF_E_NULL);
    NEF_REQUIRE(P > 0 && Bp > 0 && C > 0 && L > 0, NEF_E_SHAPE);
    hipLaunchKernelGGL(affine_relu_fwd_kernel, dim3(nef_stream_grid((int64_t)P * Bp * C, 4)), dim3(256), 0, NEF_ST, x,
                       a, b, y, P, Bp, C, L);
    return nef_launch_status();
}

size_t nef_bn_bwd_ws_bytes(int P, int Bp, int C) { return nef_bn_ws_bytes(P, C) + (size_t)P * Bp * C * sizeof(double); }

static int bn_relu_bwd_impl(const float* gy, const float* x, const float* gamma, const float* mean, const float* invstd,
                            const float* a, const float* b, float* gx, float* ggamma, float* gbeta, float* gx_chan_sum, void* ws,
                            size_t ws_bytes, int P, int Bp, int C, int L, const float* slots, int nslot, int deint,
                            nef_stream_t stream) {
    NEF_ENTER();
    (void)gamma;
    NEF_REQUIRE(gy && x && mean && invstd && a && b && gx && ggamma && gbeta && ws, NEF_E_NULL);
    NEF_REQUIRE(P > 0 && Bp > 0 && C > 0 && L > 0, NEF_E_SHAPE);
    NEF_REQUIRE(!deint || ((L & 3) == 0 && (int64_t)P * Bp * C <= 0x7FFFFFFF), NEF_E_SHAPE);
    NEF_REQUIRE(!slots || (nslot > 0 && (int64_t)Bp * nslot <= 0x7FFFFFFF), NEF_E_SHAPE);
    NEF_REQUIRE(ws_bytes >= nef_bn_bwd_ws_bytes(P, Bp, C), NEF_E_WORKSPACE);
    double* part = (double*)ws;
    float* coef = (float*)((char*)ws + (size_t)P * C * BN_SPLIT * 2 * sizeof(double));
    double* rowsum = gx_chan_sum ? (double*)((char*)ws + nef_bn_ws_bytes(P, C)) : nullptr;
    if (slots) {        // the producing conv left the sums per slot: add them up in fp64, fixed order
        hipLaunchKernelGGL(bn_slots_bwd_fused, dim3((unsigned)C), dim3(BNF_THREADS), 0, NEF_ST, slots, coef, ggamma, gbeta, P, Bp, C, L, Bp * nslot);
    } else {
        hipLaunchKernelGGL(bn_bwd_partial<0>, dim3(P * C * BN_SPLIT), dim3(256), 0, NEF_ST, gy, x, mean, invstd, a, b, part,
                           P, Bp, C, L, (const float*)nullptr);
        hipLaunchKernelGGL(bn_bwd_final, dim3((C + 63) / 64), dim3(64), 0, NEF_ST, (const double*)part, coef, ggamma,
                           gbeta, P, Bp, C, L, BN_SPLIT);
    }
    const int64_t rows = (int64_t)P * Bp * C;
    if ((L & 3) == 0 && rows <= 0x7FFFFFFF)
        hipLaunchKernelGGL(bn_bwd_apply_rows<0>, dim3((unsigned)rows), dim3(256), 0, NEF_ST, gy, x, mean, invstd, a, b,
                           (const float*)coef, gx, rowsum, Bp, C, L >> 2, (const float*)nullptr, deint);
    else
        hipLaunchKernelGGL(bn_bwd_apply, dim3(nef_stream_grid(rows, 4)), dim3(256), 0, NEF_ST, gy, x, mean, invstd, a, b,
                           (const float*)coef, gx, rowsum, P, Bp, C, L);
    if (gx_chan_sum)
        hipLaunchKernelGGL(rowsum_to_channel, dim3(C), dim3(256), 0, NEF_ST, (const double*)rowsum, gx_chan_sum,
                           P * Bp, C);
    return nef_launch_status();
}

int nef_bn_relu_bwd(const float* gy, const float* x, const float* gamma, const float* mean, const float* invstd,
                    const float* a, const float* b, float* gx, float* ggamma, float* gbeta, float* gx_chan_sum, void* ws,
                    size_t ws_bytes, int P, int Bp, int C, int L, const float* slots, int nslot, nef_stream_t stream) {
    return bn_relu_bwd_impl(gy, x, gamma, mean, invstd, a, b, gx, ggamma, gbeta, gx_chan_sum, ws, ws_bytes, P, Bp, C, L, slots, nslot, 0,
                            stream);
}

int nef_bn_relu_bwd_phase_major(const float* gy, const float* x, const float* gamma, const float* mean, const float* invstd,
                                const float* a, const float* b, float* gx, float* ggamma, float* gbeta, float* gx_chan_sum,
                                void* ws, size_t ws_bytes, int P, int Bp, int C, int L, const float* slots, int nslot,
                                nef_stream_t stream) {
    return bn_relu_bwd_impl(gy, x, gamma, mean, invstd, a, b, gx, ggamma, gbeta, gx_chan_sum, ws, ws_bytes, P, Bp, C, L, slots, nslot, 1,
                            stream);
}

int nef_bn_relu_bwd_up(const float* gu, const float* x, const float* mean, const float* invstd, const float* a,
                       const float* b, float* gx, float* ggamma, float* gbeta, float* gx_chan_sum, void* ws,
                       size_t ws_bytes, int P, int Bp, int C, int L, const float* slots, int nslot, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(gu && x && mean && invstd && a && b && gx && ggamma && gbeta && ws, NEF_E_NULL);
    NEF_REQUIRE(P > 0 && Bp > 0 && C > 0 && L >= 8 && (L & 3) == 0, NEF_E_SHAPE);
    NEF_REQUIRE(!slots || (nslot > 0 && (int64_t)Bp * nslot <= 0x7FFFFFFF), NEF_E_SHAPE);
    const int64_t rows = (int64_t)P * Bp * C;
    NEF_REQUIRE(rows <= 0x7FFFFFFF, NEF_E_SHAPE);
    NEF_REQUIRE(ws_bytes >= nef_bn_bwd_ws_bytes(P, Bp, C), NEF_E_WORKSPACE);
    double* part = (double*)ws;
    float* coef = (float*)((char*)ws + (size_t)P * C * BN_SPLIT * 2 * sizeof(double));
    double* rowsum = gx_chan_sum ? (double*)((char*)ws + nef_bn_ws_bytes(P, C)) : nullptr;
    if (slots) {
        hipLaunchKernelGGL(bn_slots_bwd_fused, dim3((unsigned)C), dim3(BNF_THREADS), 0, NEF_ST, slots, coef, ggamma, gbeta, P, Bp, C, L, Bp * nslot);
    } else {
        hipLaunchKernelGGL(bn_bwd_partial<2>, dim3(P * C * BN_SPLIT), dim3(256), 0, NEF_ST, gu, x, mean, invstd, a, b, part,
                           P, Bp, C, L, (const float*)nullptr);
        hipLaunchKernelGGL(bn_bwd_final, dim3((C + 63) / 64), dim3(64), 0, NEF_ST, (const double*)part, coef, ggamma,
                           gbeta, P, Bp, C, L, BN_SPLIT);
    }
    hipLaunchKernelGGL(bn_bwd_apply_rows<2>, dim3((unsigned)rows), dim3(256), 0, NEF_ST, gu, x, mean, invstd, a, b,
                       (const float*)coef, gx, rowsum, Bp, C, L >> 2, (const float*)nullptr);
    if (gx_chan_sum)
        hipLaunchKernelGGL(rowsum_to_channel, dim3(C), dim3(256), 0, NEF_ST, (const double*)rowsum, gx_chan_sum,
                           P * Bp, C);
    return nef_launch_status();
}

static int bn_relu_bwd_combine3_impl(const float* gy, const float* x, const float* mean, const float* invstd, const float* a,
                                     const float* b, float* gP2, float* ggamma, float* gbeta, float* gx_chan_sum, void* ws,
                                     size_t ws_bytes, int Bp, int C, int L, const float* slots, int nslot, int deint,
                                     nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(!deint || (L & 1) == 0, NEF_E_SHAPE);
    NEF_REQUIRE(gy && x && mean && invstd && a && b && gP2 && ggamma && gbeta && ws, NEF_E_NULL);
    NEF_REQUIRE(Bp > 0 && C > 0 && L > 0 && (int64_t)Bp * C <= 0x7FFFFFFF, NEF_E_SHAPE);
    NEF_REQUIRE(!slots || (nslot > 0 && (int64_t)Bp * nslot <= 0x7FFFFFFF), NEF_E_SHAPE);
    NEF_REQUIRE(ws_bytes >= nef_bn_bwd_ws_bytes(3, Bp, C), NEF_E_WORKSPACE);
    double* part = (double*)ws;
    float* coef = (float*)((char*)ws + (size_t)3 * C * BN_SPLIT * 2 * sizeof(double));
    double* rowsum = gx_chan_sum ? (double*)((char*)ws + nef_bn_ws_bytes(3, C)) : nullptr;
    if (slots) {
        hipLaunchKernelGGL(bn_slots_bwd_fused, dim3((unsigned)C), dim3(BNF_THREADS), 0, NEF_ST, slots, coef, ggamma, gbeta, 3, Bp, C, L, Bp * nslot);
    } else {
        hipLaunchKernelGGL(bn_bwd_partial<0>, dim3(3 * C * BN_SPLIT), dim3(256), 0, NEF_ST, gy, x, mean, invstd, a, b, part,
                           3, Bp, C, L, (const float*)nullptr);
        hipLaunchKernelGGL(bn_bwd_final, dim3((C + 63) / 64), dim3(64), 0, NEF_ST, (const double*)part, coef, ggamma,
                           gbeta, 3, Bp, C, L, BN_SPLIT);
    }
    hipLaunchKernelGGL(bn_bwd_apply_combine3, dim3((unsigned)(Bp * C)), dim3(256), 0, NEF_ST, gy, x, mean, invstd, a, b,
                       (const float*)coef, gP2, rowsum, Bp, C, L, deint);
    if (gx_chan_sum)
        hipLaunchKernelGGL(rowsum_to_channel, dim3(C), dim3(256), 0, NEF_ST, (const double*)rowsum, gx_chan_sum, 3 * Bp,
                           C);
    return nef_launch_status();
}

int nef_bn_relu_bwd_combine3(const float* gy, const float* x, const float* mean, const float* invstd, const float* a,
                             const float* b, float* gP2, float* ggamma, float* gbeta, float* gx_chan_sum, void* ws,
                             size_t ws_bytes, int Bp, int C, int L, const float* slots, int nslot, nef_stream_t stream) {
    return bn_relu_bwd_combine3_impl(gy, x, mean, invstd, a, b, gP2, ggamma, gbeta, gx_chan_sum, ws, ws_bytes, Bp, C, L, slots, nslot, 0,
                                     stream);
}

int nef_bn_relu_bwd_combine3_phase_major(const float* gy, const float* x, const float* mean, const float* invstd, const float* a,
                                         const float* b, float* gP2, float* ggamma, float* gbeta, float* gx_chan_sum, void* ws,
                                         size_t ws_bytes, int Bp, int C, int L, const float* slots, int nslot,
                                         nef_stream_t stream) {
    return bn_relu_bwd_combine3_impl(gy, x, mean, invstd, a, b, gP2, ggamma, gbeta, gx_chan_sum, ws, ws_bytes, Bp, C, L, slots, nslot, 1,
                                     stream);
}

size_t nef_bn_bwd_outconv_ws_bytes(int P, int Bp, int C, int L) {
    return nef_bn_bwd_ws_bytes(P, Bp, C) + (size_t)P * Bp * L * sizeof(float);
}

int nef_bn_relu_bwd_outconv(const float* gout, const float* out, const float* wout, const float* x, const float* mean,
                            const float* invstd, const float* a, const float* b, float* gx, float* ggamma, float* gbeta,
                            float* gx_chan_sum, void* ws, size_t ws_bytes, int P, int Bp, int C, int L,
                            nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(gout && out && wout && x && mean && invstd && a && b && gx && ggamma && gbeta && ws, NEF_E_NULL);
    NEF_REQUIRE(P > 0 && Bp > 0 && C > 0 && L > 0 && (L & 3) == 0, NEF_E_SHAPE);
    const int64_t rows = (int64_t)P * Bp * C;
    NEF_REQUIRE(rows <= 0x7FFFFFFF, NEF_E_SHAPE);
    NEF_REQUIRE(ws_bytes >= nef_bn_bwd_outconv_ws_bytes(P, Bp, C, L), NEF_E_WORKSPACE);
    double* part = (double*)ws;
    float* coef = (float*)((char*)ws + (size_t)P * C * BN_SPLIT * 2 * sizeof(double));
    double* rowsum = gx_chan_sum ? (double*)((char*)ws + nef_bn_ws_bytes(P, C)) : nullptr;
    float* go = (float*)((char*)ws + nef_bn_bwd_ws_bytes(P, Bp, C));
    const int64_t n = (int64_t)P * Bp * L;
    hipLaunchKernelGGL(outconv_go_kernel, dim3(nef_stream_grid(n, 256)), dim3(256), 0, NEF_ST, gout, out, go, n);
    hipLaunchKernelGGL(bn_bwd_partial<1>, dim3(P * C * BN_SPLIT), dim3(256), 0, NEF_ST, (const float*)go, x, mean, invstd,
                       a, b, part, P, Bp, C, L, wout);
    hipLaunchKernelGGL(bn_bwd_final, dim3((C + 63) / 64), dim3(64), 0, NEF_ST, (const double*)part, coef, ggamma, gbeta,
                       P, Bp, C, L);
    hipLaunchKernelGGL(bn_bwd_apply_rows<1>, dim3((unsigned)rows), dim3(256), 0, NEF_ST, (const float*)go, x, mean,
                       invstd, a, b, (const float*)coef, gx, rowsum, Bp, C, L >> 2, wout);
    if (gx_chan_sum)
        hipLaunchKernelGGL(rowsum_to_channel, dim3(C), dim3(256), 0, NEF_ST, (const double*)rowsum, gx_chan_sum,
                           P * Bp, C);
    return nef_launch_status();
}

int nef_outconv_fwd_pro(const float* x, const float* a, const float* b, int Bp, const float* w, const float* bias,
                        float* out, int N, int C, int L, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(x && w && bias && out, NEF_E_NULL);
    NEF_REQUIRE((a == nullptr) == (b == nullptr), NEF_E_NULL);
    NEF_REQUIRE(N > 0 && C > 0 && C <= OC_MAXC && L > 0 && (!a || Bp > 0), NEF_E_SHAPE);
    if ((L & 3) == 0) {
        const int tiles = (L + 1023) / 1024;
        hipLaunchKernelGGL(outconv_fwd_kernel<true>, dim3((unsigned)((int64_t)N * tiles)), dim3(256), 0, NEF_ST, x, w, bias,
                           out, N, C, L, tiles, a, b, a ? Bp : 1);
        return nef_launch_status();
    }
    const int tiles = (L + 255) / 256;
    hipLaunchKernelGGL(outconv_fwd_kernel<false>, dim3((unsigned)((int64_t)N * tiles)), dim3(256), 0, NEF_ST, x, w, bias,
                       out, N, C, L, tiles, a, b, a ? Bp : 1);
    return nef_launch_status();
}

int nef_outconv_fwd(const float* x, const float* w, const float* bias, float* out, int N, int C, int L,
                    nef_stream_t stream) {
    return nef_outconv_fwd_pro(x, nullptr, nullptr, 1, w, bias, out, N, C, L, stream);
}

int nef_outconv_bwd_data(const float* gout, const float* out, const float* w, float* gx, int N, int C, int L,
                         nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(gout && out && w && gx, NEF_E_NULL);
    NEF_REQUIRE(N > 0 && C > 0 && C <= OC_MAXC && L > 0, NEF_E_SHAPE);
    const int tiles = (L + 255) / 256;
    hipLaunchKernelGGL(outconv_bwd_data_kernel, dim3((unsigned)((int64_t)N * tiles)), dim3(256), 0, NEF_ST, gout, out,
                       w, gx, N, C, L, tiles);
    return nef_launch_status();
}

size_t nef_outconv_bwd_weight_ws_bytes(int C) { return (size_t)OC_BLOCKS * (C + 1) * 3 * sizeof(double); }

int nef_outconv_bwd_weight_pro(const float* gout, const float* out, const float* x, const float* a, const float* b, int Bp,
                               float* gw, float* gb, void* ws, size_t ws_bytes, int N, int C, int L, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(gout && out && x && gw && gb && ws, NEF_E_NULL);
    NEF_REQUIRE((a == nullptr) == (b == nullptr), NEF_E_NULL);
    NEF_REQUIRE(N > 0 && C > 0 && C <= OC_MAXC && L > 0 && (!a || Bp > 0), NEF_E_SHAPE);
    NEF_REQUIRE(ws_bytes >= nef_outconv_bwd_weight_ws_bytes(C), NEF_E_WORKSPACE);
    const int tiles = (L + OC_TILE - 1) / OC_TILE;
    const int64_t units = (int64_t)N * tiles;
    const int nblk = (int)(units < OC_BLOCKS ? units : OC_BLOCKS);
    hipLaunchKernelGGL(outconv_bwd_weight_partial, dim3(nblk), dim3(256), 0, NEF_ST, gout, out, x, (double*)ws, N, C, L,
                       tiles, a, b, a ? Bp : 1);
    hipLaunchKernelGGL(outconv_bwd_weight_final, dim3((C + 1) * 3), dim3(256), 0, NEF_ST, (const double*)ws, gw, gb, C,
                       nblk);
    return nef_launch_status();
}

int nef_outconv_bwd_weight(const float* gout, const float* out, const float* x, float* gw, float* gb, void* ws,
                           size_t ws_bytes, int N, int C, int L, nef_stream_t stream) {
    return nef_outconv_bwd_weight_pro(gout, out, x, nullptr, nullptr, 1, gw, gb, ws, ws_bytes, N, C, L, stream);
}

size_t nef_loss_ws_bytes(void) { return (size_t)LOSS_BLOCKS * 3 * sizeof(double); }

int nef_loss_fwd(const float* pred, const float* pred_p, const float* pred_l, const float* target, float* losses,
                 void* ws, size_t ws_bytes, int64_t n, float f0, float f1, float f2, int reg_l2, int use_mask,
                 nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(pred && pred_p && pred_l && target && losses && ws, NEF_E_NULL);
    NEF_REQUIRE(n > 0, NEF_E_SHAPE);
    NEF_REQUIRE(ws_bytes >= nef_loss_ws_bytes(), NEF_E_WORKSPACE);
    hipLaunchKernelGGL(loss_partial, dim3(LOSS_BLOCKS), dim3(256), 0, NEF_ST, pred, pred_p, pred_l, target, (double*)ws,
                       n, reg_l2);
    hipLaunchKernelGGL(loss_final, dim3(1), dim3(256), 0, NEF_ST, (const double*)ws, losses, LOSS_BLOCKS, n, f0, f1, f2,
                       use_mask);
    return nef_launch_status();
}

int nef_loss_bwd(const float* pred, const float* pred_p, const float* pred_l, const float* target, const float* gscale,
                 float* g_pred, float* g_p, float* g_l, int64_t n, float f0, float f1, float f2, int reg_l2,
                 int use_mask, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(pred && pred_p && pred_l && target && g_pred && g_p && g_l, NEF_E_NULL);
    NEF_REQUIRE(n > 0, NEF_E_SHAPE);
    hipLaunchKernelGGL(loss_bwd_kernel, dim3(nef_stream_grid(n, 256)), dim3(256), 0, NEF_ST, pred, pred_p, pred_l,
                       target, gscale, g_pred, g_p, g_l, n, f0, f1, f2, reg_l2, use_mask);
    return nef_launch_status();
}

int nef_sgd_momentum(float* p, const float* g, float* buf, int64_t n, float lr, float mu, float gscale, int first_step,
                     const float* skip_if_positive, int32_t* skipped, const float* lr_dev, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(p && g && buf, NEF_E_NULL);
    NEF_REQUIRE(n > 0, NEF_E_SHAPE);
    hipLaunchKernelGGL(sgd_kernel, dim3(nef_stream_grid(n, 256)), dim3(256), 0, NEF_ST, p, g, buf, n, lr, mu, gscale,
                       first_step, skip_if_positive, skipped, lr_dev);
    return nef_launch_status();
}

int nef_slots_to_rows(const float* slots, int nslot, float* out, int B, int C, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(slots && out, NEF_E_NULL);
    NEF_REQUIRE(nslot > 0 && B > 0 && C > 0, NEF_E_SHAPE);
    const int64_t n = (int64_t)B * C;
    hipLaunchKernelGGL(slots_to_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, NEF_ST, slots, nslot, out, B, C);
    return nef_launch_status();
}

int nef_poly_weights(const float* w, float* wsyn, int rows, int Cig, int tile_Cr, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(w && wsyn, NEF_E_NULL);
    NEF_REQUIRE(rows > 0 && Cig > 0 && (tile_Cr == 0 || (tile_Cr > 0 && tile_Cr % 64 == 0 && rows % tile_Cr == 0)), NEF_E_SHAPE);
    const int64_t n = (int64_t)rows * Cig;
    hipLaunchKernelGGL(poly_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, NEF_ST, w, wsyn, rows, Cig, tile_Cr);
    return nef_launch_status();
}

size_t nef_poly_wgrad_fold_ws_bytes(int B, int G, int Cog, int Cig) {
    return (size_t)((B + POLY_BC - 1) / POLY_BC) * G * Cog * Cig * 2 * sizeof(float);
}

int nef_poly_wgrad_fold(const float* gw2, const float* gy_pm, const float* xedge, float* gw, void* ws, size_t ws_bytes, int B, int G,
                        int Cog, int Cig, int T, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(gw2 && gy_pm && xedge && gw && ws, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && G > 0 && Cog > 0 && Cig > 0 && T >= 4 && T % 2 == 0, NEF_E_SHAPE);
    const int chunks = (B + POLY_BC - 1) / POLY_BC;
    NEF_REQUIRE(chunks <= 65535, NEF_E_SHAPE);
    NEF_REQUIRE(ws_bytes >= nef_poly_wgrad_fold_ws_bytes(B, G, Cog, Cig), NEF_E_WORKSPACE);
    hipLaunchKernelGGL(poly_wgrad_edge_partial, dim3((unsigned)(G * Cog), (unsigned)chunks), dim3(256), 0, NEF_ST, gy_pm, xedge,
                       (float*)ws, B, G, Cog, Cig, T / 2);
    hipLaunchKernelGGL(poly_wgrad_fold_kernel, dim3((unsigned)(G * Cog)), dim3(256), 0, NEF_ST, gw2, (const float*)ws, gw, G * Cog, Cig,
                       chunks);
    return nef_launch_status();
}

int nef_poly_fwd_edge(const float* x, const float* w, float* y, int B, int G, int Cr, int Cig, int T, const float* pro_a,
                      const float* pro_b, int pro_Bp, float* stats, int nslot, float* xedge, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(x && w && y, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && G > 0 && Cr > 0 && Cig > 0 && Cig <= 4096 && T >= 4 && T % 2 == 0, NEF_E_SHAPE);
    NEF_REQUIRE((!pro_a && !pro_b) || (pro_a && pro_b && pro_Bp > 0), NEF_E_NULL);
    NEF_REQUIRE(!stats || nslot > 0, NEF_E_SHAPE);
    hipLaunchKernelGGL(poly_fwd_edge_kernel, dim3((unsigned)(B * G)), dim3(256), (size_t)2 * Cig * sizeof(float), NEF_ST, x, w, y, B, G, Cr,
                       Cig, T, pro_a, pro_b, pro_Bp, stats, nslot, xedge);
    return nef_launch_status();
}

int nef_poly_bwd_edge(const float* gy, const float* w, float* gx, int B, int G, int Cog, int Cig, int T, const float* bnb_x,
                      const float* bnb_mean, const float* bnb_invstd, const float* bnb_a, const float* bnb_b, int bnb_Bp,
                      float* bnb_slots, int nslot, int gy_phase_major, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(gy && w && gx, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && G > 0 && Cog > 0 && Cig > 0 && T >= 4 && T % 2 == 0 && Cog <= 4096, NEF_E_SHAPE);
    NEF_REQUIRE(!bnb_slots || (bnb_x && bnb_mean && bnb_invstd && bnb_a && bnb_b && bnb_Bp > 0 && nslot > 0), NEF_E_NULL);
    hipLaunchKernelGGL(poly_bwd_edge_kernel, dim3((unsigned)(B * G)), dim3(256), (size_t)4 * Cog * sizeof(float), NEF_ST, gy, w, gx, B, G,
                       Cog, Cig, T, bnb_x, bnb_mean, bnb_invstd, bnb_a, bnb_b, bnb_Bp, bnb_slots, nslot, gy_phase_major);
    return nef_launch_status();
}

int nef_step_words(int32_t* choice, int64_t* seed, int c1, int c2, int64_t seed_value, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(choice && seed, NEF_E_NULL);
    hipLaunchKernelGGL(step_words_kernel, dim3(1), dim3(64), 0, NEF_ST, choice, seed, c1, c2, seed_value);
    return nef_launch_status();
}

int nef_amax_roll(float* cur, float* nxt, int n, float follow_up, float follow_down, int follow_always, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(cur && nxt, NEF_E_NULL);
    NEF_REQUIRE(n >= 0 && follow_up >= 1.f && follow_down >= 1.f, NEF_E_SHAPE);
    if (n == 0) return NEF_OK;
    hipLaunchKernelGGL(amax_roll_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, NEF_ST, cur, nxt, n, follow_up, follow_down,
                       follow_always);
    return nef_launch_status();
}

int nef_flatten(const float* const* srcs, const int64_t* sizes, int n, float* out, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE((srcs && sizes && out) || n == 0, NEF_E_NULL);
    NEF_REQUIRE(n >= 0, NEF_E_SHAPE);
    int64_t off = 0;
    for (int i0 = 0; i0 < n; i0 += FLAT_MAX) {
        FlatTable t;
        const int m = n - i0 < FLAT_MAX ? n - i0 : FLAT_MAX;
        int64_t biggest = 1;
        for (int i = 0; i < m; ++i) {
            NEF_REQUIRE(srcs[i0 + i] && sizes[i0 + i] >= 0, NEF_E_NULL);
            t.d[i] = FlatDesc{srcs[i0 + i], sizes[i0 + i], off};
            off += sizes[i0 + i];
            if (sizes[i0 + i] > biggest) biggest = sizes[i0 + i];
        }
        int gx = (int)nef_cdiv(biggest, 256 * 4 * 4);      // ~4 16-byte transfers per thread on the largest tensor
        if (gx > 128) gx = 128;
        if (gx < 1) gx = 1;
        hipLaunchKernelGGL(flatten_kernel, dim3((unsigned)gx, (unsigned)m), dim3(256), 0, NEF_ST, t, out);
    }
    return nef_launch_status();
}

int nef_regroup_halves(const float* src, float* dst, int Co, int Cih, int K, int inverse, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(src && dst, NEF_E_NULL);
    NEF_REQUIRE(Co > 0 && Cih > 0 && K > 0, NEF_E_SHAPE);
    const int64_t n = (int64_t)2 * Co * Cih * K;
    hipLaunchKernelGGL(regroup_halves_kernel, dim3((unsigned)nef_stream_grid(n, 256)), dim3(256), 0, NEF_ST, src, dst, Co, Cih, K, inverse);
    return nef_launch_status();
}

// diagnostics (not in the header): `wgs` workgroups busy for `us` microseconds on `stream` (bench.py --dry-collective)
int nef_debug_spin_us(float us, int wgs, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(us >= 0.f && us <= 50000.f && wgs >= 1 && wgs <= 64, NEF_E_SHAPE);
    int dev = 0, khz = 100000;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
    const unsigned long long ticks = (unsigned long long)((double)us * 1e-3 * (double)khz);
    hipLaunchKernelGGL(debug_spin_kernel, dim3((unsigned)wgs), dim3(64), 0, NEF_ST, ticks);
    return nef_launch_status();
}

int nef_h2_taint(const int32_t* clamped_total, int32_t* mark, float* out, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(clamped_total && mark && out, NEF_E_NULL);
    hipLaunchKernelGGL(h2_taint_kernel, dim3(1), dim3(1), 0, NEF_ST, clamped_total, mark, out);
    return nef_launch_status();
}

}  // extern "C"
