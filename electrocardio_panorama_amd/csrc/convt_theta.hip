// (1) ConvTranspose1d(k=2, s=2, groups=G, bias) on the fixed-length ROI latents -- reference
//     codes/network/model_nefnet.py:96-97 (z2_conv2[1]): x [B][G*Cig][T] -> y [B][G*Cog][2T] with
//     y[b][g*Cog+co][2t+j] = bias + sum_ci x[b][g*Cig+ci][t] * w[g*Cig+ci][co][j].  With stride == kernel the op
//     is two interleaved 1x1 convolutions; T is 16, so a workgroup handles one (sample-group, conv group) and
//     keeps the [Cig][Cog][2] filter slab in LDS.  <2 % of the step's FLOPs: plain fp32 FMA, no MFMA.
// (2) Angular encoding + Linear(12->O) -- codes/network/utils/theta_encoder.py:13-29 and
//     model_nefnet.py:76-77,121,164,183.
#include "nef_common.h"

namespace {

constexpr int CT_CIG = 128, CT_COG = 64;
constexpr int CT_NB = 4;   // samples per workgroup

// LDS: wl [Cig][Cog*2] (64 KB) + xl [NB][Cig][T<=32]
__global__ __launch_bounds__(256) void convt2_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int B,
                                                         int G, int T) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem;                               // [Cig][Cog][2]
    float* xl = smem + CT_CIG * CT_COG * 2;         // [NB][Cig][T]
    const int g = blockIdx.x % G;
    const int b0 = (blockIdx.x / G) * CT_NB;
    for (int i = threadIdx.x; i < CT_CIG * CT_COG * 2; i += 256) wl[i] = w[(int64_t)g * CT_CIG * CT_COG * 2 + i];
    for (int i = threadIdx.x; i < CT_NB * CT_CIG * T; i += 256) {
        const int nb = i / (CT_CIG * T);
        const int r = i - nb * CT_CIG * T;
        const int b = b0 + nb;
        xl[i] = b < B ? x[((int64_t)b * G + g) * CT_CIG * T + r] : 0.f;
    }
    __syncthreads();
    const int To = 2 * T;
    // outputs per workgroup: NB * Cog * To
    for (int o = threadIdx.x; o < CT_NB * CT_COG * To; o += 256) {
        const int u = o % To;
        const int co = (o / To) % CT_COG;
        const int nb = o / (To * CT_COG);
        const int b = b0 + nb;
        if (b >= B) continue;
        const int t = u >> 1, j = u & 1;
        const float* xs = xl + nb * CT_CIG * T + t;
        const float* ws = wl + co * 2 + j;
        float acc = 0.f;
#pragma unroll 8
        for (int ci = 0; ci < CT_CIG; ++ci) acc = fmaf(xs[ci * T], ws[ci * CT_COG * 2], acc);
        y[(((int64_t)b * G + g) * CT_COG + co) * To + u] = acc + bias[g * CT_COG + co];
    }
}

// gx[b][g*Cig+ci][t] = sum_{co,j} gy[b][g*Cog+co][2t+j] * w[g*Cig+ci][co][j]
__global__ __launch_bounds__(256) void convt2_bwd_data_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                              float* __restrict__ gx, int B, int G, int T) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem;                               // [Cig][Cog*2 + 1]  (padded: lanes walk ci)
    float* gl = smem + CT_CIG * (CT_COG * 2 + 1);   // [NB][Cog][2T]
    const int To = 2 * T;
    const int g = blockIdx.x % G;
    const int b0 = (blockIdx.x / G) * CT_NB;
    for (int i = threadIdx.x; i < CT_CIG * CT_COG * 2; i += 256) {
        const int ci = i / (CT_COG * 2), r = i % (CT_COG * 2);
        wl[ci * (CT_COG * 2 + 1) + r] = w[(int64_t)g * CT_CIG * CT_COG * 2 + i];
    }
    for (int i = threadIdx.x; i < CT_NB * CT_COG * To; i += 256) {
        const int nb = i / (CT_COG * To);
        const int r = i - nb * CT_COG * To;
        const int b = b0 + nb;
        gl[i] = b < B ? gy[((int64_t)b * G + g) * CT_COG * To + r] : 0.f;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < CT_NB * CT_CIG * T; o += 256) {
        const int t = o % T;
        const int ci = (o / T) % CT_CIG;
        const int nb = o / (T * CT_CIG);
        const int b = b0 + nb;
        if (b >= B) continue;
        const float* gs = gl + nb * CT_COG * To + 2 * t;
        const float* ws = wl + ci * (CT_COG * 2 + 1);
        float acc = 0.f;
#pragma unroll 8
        for (int co = 0; co < CT_COG; ++co) {
            acc = fmaf(gs[co * To], ws[co * 2], acc);
            acc = fmaf(gs[co * To + 1], ws[co * 2 + 1], acc);
        }
        gx[(((int64_t)b * G + g) * CT_CIG + ci) * T + t] = acc;
    }
}

// gw[g*Cig+ci][co][j] = sum_{b,t} x[b][g*Cig+ci][t] * gy[b][g*Cog+co][2t+j]; partials over sample splits.
constexpr int CT_SPLIT = 32;
__global__ __launch_bounds__(256) void convt2_bwd_weight_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ gy, float* __restrict__ part,
                                                                int B, int G, int T) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int To = 2 * T;
    float* xl = smem;                    // [Cig][T+1]
    float* gl = smem + CT_CIG * (T + 1); // [Cog][2T+1]
    const int g = blockIdx.x % G;
    const int sp = blockIdx.x / G;
    // thread owns outputs o = threadIdx.x + 256*q, q < 64  (Cig*Cog*2 = 16384); o = (ci*Cog + co)*2 + j
    float acc[64];
#pragma unroll
    for (int q = 0; q < 64; ++q) acc[q] = 0.f;
    for (int b = sp; b < B; b += CT_SPLIT) {
        __syncthreads();
        for (int i = threadIdx.x; i < CT_CIG * T; i += 256)
            xl[(i / T) * (T + 1) + (i % T)] = x[((int64_t)b * G + g) * CT_CIG * T + i];
        for (int i = threadIdx.x; i < CT_COG * To; i += 256)
            gl[(i / To) * (To + 1) + (i % To)] = gy[((int64_t)b * G + g) * CT_COG * To + i];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 64; ++q) {
            const int o = threadIdx.x + 256 * q;
            const int j = o & 1;
            const int co = (o >> 1) % CT_COG;
            const int ci = (o >> 1) / CT_COG;
            const float* xs = xl + ci * (T + 1);
            const float* gs = gl + co * (To + 1) + j;
            float a = acc[q];
            for (int t = 0; t < T; ++t) a = fmaf(xs[t], gs[2 * t], a);
            acc[q] = a;
        }
    }
#pragma unroll
    for (int q = 0; q < 64; ++q)
        part[((int64_t)sp * G + g) * CT_CIG * CT_COG * 2 + threadIdx.x + 256 * q] = acc[q];
}

__global__ void convt2_bwd_weight_reduce(const float* __restrict__ part, float* __restrict__ gw, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int sp = 0; sp < CT_SPLIT; ++sp) s += part[(int64_t)sp * n + i];
        gw[i] = s;
    }
}


// ---------------- ConvTranspose1d(k=2,s=2) as a grouped 1x1 conv on the matrix cores ----------------
// y[b][g][co][2t+j] = bias + sum_ci x[b][g][ci][t] * w[g][ci][co][j] is a 1x1 conv onto 2*Cog "channels" m = co*2+j
// followed by an interleave of (m, t) -> (co, 2t+j).  The three helpers below are the layout passes around
// nef_conv_fwd / nef_conv_bwd_weight with K = 1 (a few tens of MB per step).
// per-group transpose: out[g][c][r] = in[g][r][c]
__global__ void group_transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int G, int R, int Cn) {
    __shared__ float tile[32][33];
    const int g = blockIdx.z, r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const float* ig = in + (size_t)g * R * Cn;
    float* og = out + (size_t)g * R * Cn;
    for (int i = ly; i < 32; i += 8)
        if (r0 + i < R && c0 + lx < Cn) tile[i][lx] = ig[(size_t)(r0 + i) * Cn + c0 + lx];
    __syncthreads();
    for (int i = ly; i < 32; i += 8)
        if (c0 + i < Cn && r0 + lx < R) og[(size_t)(c0 + i) * R + r0 + lx] = tile[lx][i];
}

// y[n][c][2t+j] = yq[n][2c+j][t] + bias[c]     (rows n = b*G*Cog-flattened: C = total output channels)
__global__ void convt_interleave_kernel(const float* __restrict__ yq, const float* __restrict__ bias,
                                        float* __restrict__ y, int64_t rows, int C, int T) {
    const int64_t total = rows * 2 * T;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int u = (int)(i % (2 * T));
        const int64_t row = i / (2 * T);                   // (b, c)
        const int t = u >> 1, j = u & 1;
        const float bv = bias ? bias[row % C] : 0.f;
        y[i] = yq[(row * 2 + j) * T + t] + bv;
    }
}

// gyq[n][2c+j][t] = gy[n][c][2t+j]
__global__ void convt_deinterleave_kernel(const float* __restrict__ gy, float* __restrict__ gyq, int64_t rows, int T) {
    const int64_t total = rows * 2 * T;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % T);
        const int64_t rj = i / T;                          // row*2 + j
        gyq[i] = gy[(rj >> 1) * 2 * T + 2 * t + (rj & 1)];
    }
}

// ---------------- angular encoding ----------------
__device__ __forceinline__ void encode12(float th, float ph, float (&e)[12]) {
    const float a[4] = {th, ph, th + ph, th - ph};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        e[3 * i] = a[i];
        e[3 * i + 1] = sinf(a[i]);
        e[3 * i + 2] = cosf(a[i]);
    }
}

__device__ __forceinline__ float enc_component(float th, float ph, int k) {
    const int i = k / 3, f = k % 3;
    const float a = i == 0 ? th : (i == 1 ? ph : (i == 2 ? th + ph : th - ph));
    return f == 0 ? a : (f == 1 ? sinf(a) : cosf(a));
}

__global__ void theta_encode_kernel(const float* __restrict__ theta, float* __restrict__ enc, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float e[12];
    encode12(theta[2 * n], theta[2 * n + 1], e);
#pragma unroll
    for (int i = 0; i < 12; ++i) enc[n * 12 + i] = e[i];
}

__global__ void theta_mlp_fwd_kernel(const float* __restrict__ theta, const float* __restrict__ W,
                                     const float* __restrict__ bias, float* __restrict__ y, int N, int O) {
    const int64_t total = (int64_t)N * O;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / O), o = (int)(i % O);
        float e[12];
        encode12(theta[2 * n], theta[2 * n + 1], e);
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 12; ++k) acc = fmaf(e[k], W[o * 12 + k], acc);
        y[i] = acc + bias[o];
    }
}

// one workgroup per output unit o: threads stride over the N rows, each evaluating the 12-value encoding once and
// accumulating the 12 weight-gradient partials + the bias partial; fixed-order block reduction (fp64).
__global__ __launch_bounds__(256) void theta_mlp_bwd_kernel(const float* __restrict__ theta, const float* __restrict__ gy,
                                                            float* __restrict__ gW, float* __restrict__ gb, int N, int O) {
    __shared__ double sm[4];
    const int o = blockIdx.x;
    double acc[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) acc[k] = 0.0;
    for (int n = threadIdx.x; n < N; n += 256) {
        float e[12];
        encode12(theta[2 * n], theta[2 * n + 1], e);
        const float g = gy[(int64_t)n * O + o];
#pragma unroll
        for (int k = 0; k < 12; ++k) acc[k] += (double)(g * e[k]);
        acc[12] += (double)g;
    }
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        const double s = nef_block_sum_d(acc[k], sm);
        if (threadIdx.x == 0) {
            if (k < 12) gW[o * 12 + k] = (float)s;
            else gb[o] = (float)s;
        }
    }
}

}  // namespace

#define NEF_ST ((hipStream_t)stream)

extern "C" {

int nef_convt2_fwd(const float* x, const float* w, const float* bias, float* y, int B, int G, int Cig, int Cog, int T,
                   nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(x && w && bias && y, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && G > 0 && T > 0 && T <= 32, NEF_E_SHAPE);
    NEF_REQUIRE(Cig == CT_CIG && Cog == CT_COG, NEF_E_UNSUPPORTED);
    const size_t lds = (size_t)(CT_CIG * CT_COG * 2 + CT_NB * CT_CIG * T) * sizeof(float);
    static unsigned long long lds_set = 0;      // per-device bits, see nef_ensure_dyn_lds
    if (int e = nef_ensure_dyn_lds(reinterpret_cast<const void*>(&convt2_fwd_kernel), 160 * 1024, &lds_set)) return e;
    const int nbg = (B + CT_NB - 1) / CT_NB;
    hipLaunchKernelGGL(convt2_fwd_kernel, dim3(nbg * G), dim3(256), lds, NEF_ST, x, w, bias, y, B, G, T);
    return nef_launch_status();
}

int nef_convt2_bwd_data(const float* gy, const float* w, float* gx, int B, int G, int Cig, int Cog, int T,
                        nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(gy && w && gx, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && G > 0 && T > 0 && T <= 32, NEF_E_SHAPE);
    NEF_REQUIRE(Cig == CT_CIG && Cog == CT_COG, NEF_E_UNSUPPORTED);
    const size_t lds = (size_t)(CT_CIG * (CT_COG * 2 + 1) + CT_NB * CT_COG * 2 * T) * sizeof(float);
    static unsigned long long lds_set = 0;      // per-device bits, see nef_ensure_dyn_lds
    if (int e = nef_ensure_dyn_lds(reinterpret_cast<const void*>(&convt2_bwd_data_kernel), 160 * 1024, &lds_set)) return e;
    const int nbg = (B + CT_NB - 1) / CT_NB;
    hipLaunchKernelGGL(convt2_bwd_data_kernel, dim3(nbg * G), dim3(256), lds, NEF_ST, gy, w, gx, B, G, T);
    return nef_launch_status();
}

size_t nef_convt2_bwd_weight_ws_bytes(int G, int Cig, int Cog) {
    return (size_t)CT_SPLIT * G * Cig * Cog * 2 * sizeof(float) + nef_chan_sum_ws_bytes(G * Cog);
}

int nef_convt2_bwd_weight(const float* x, const float* gy, float* gw, float* gb, void* ws, size_t ws_bytes, int B,
                          int G, int Cig, int Cog, int T, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(x && gy && gw && gb && ws, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && G > 0 && T > 0 && T <= 32, NEF_E_SHAPE);
    NEF_REQUIRE(Cig == CT_CIG && Cog == CT_COG, NEF_E_UNSUPPORTED);
    NEF_REQUIRE(ws_bytes >= nef_convt2_bwd_weight_ws_bytes(G, Cig, Cog), NEF_E_WORKSPACE);
    const size_t lds = (size_t)(CT_CIG * (T + 1) + CT_COG * (2 * T + 1)) * sizeof(float);
    float* part = (float*)ws;
    hipLaunchKernelGGL(convt2_bwd_weight_kernel, dim3(G * CT_SPLIT), dim3(256), lds, NEF_ST, x, gy, part, B, G, T);
    const int64_t n = (int64_t)G * Cig * Cog * 2;
    hipLaunchKernelGGL(convt2_bwd_weight_reduce, dim3(nef_stream_grid(n, 256)), dim3(256), 0, NEF_ST,
                       (const float*)part, gw, n);
    int rc = nef_launch_status();
    if (rc != NEF_OK) return rc;
    void* cs_ws = (char*)ws + (size_t)CT_SPLIT * n * sizeof(float);
    return nef_chan_sum(gy, gb, cs_ws, nef_chan_sum_ws_bytes(G * Cog), B, G * Cog, 2 * T, stream);
}

int nef_group_transpose(const float* in, float* out, int G, int R, int Cn, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(in && out, NEF_E_NULL);
    NEF_REQUIRE(G > 0 && G <= 65535 && R > 0 && Cn > 0, NEF_E_SHAPE);
    hipLaunchKernelGGL(group_transpose_kernel, dim3((Cn + 31) / 32, (R + 31) / 32, G), dim3(256), 0, NEF_ST, in, out, G, R,
                       Cn);
    return nef_launch_status();
}

int nef_convt2_interleave(const float* yq, const float* bias, float* y, int B, int C, int T, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(yq && y, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && C > 0 && T > 0, NEF_E_SHAPE);
    const int64_t rows = (int64_t)B * C;
    hipLaunchKernelGGL(convt_interleave_kernel, dim3(nef_stream_grid(rows * 2 * T, 256)), dim3(256), 0, NEF_ST, yq, bias,
                       y, rows, C, T);
    return nef_launch_status();
}

int nef_convt2_deinterleave(const float* gy, float* gyq, int B, int C, int T, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(gy && gyq, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && C > 0 && T > 0, NEF_E_SHAPE);
    const int64_t rows = (int64_t)B * C;
    hipLaunchKernelGGL(convt_deinterleave_kernel, dim3(nef_stream_grid(rows * 2 * T, 256)), dim3(256), 0, NEF_ST, gy, gyq,
                       rows, T);
    return nef_launch_status();
}

int nef_theta_encode(const float* theta, float* enc, int N, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(theta && enc, NEF_E_NULL);
    NEF_REQUIRE(N > 0, NEF_E_SHAPE);
    hipLaunchKernelGGL(theta_encode_kernel, dim3((N + 255) / 256), dim3(256), 0, NEF_ST, theta, enc, N);
    return nef_launch_status();
}

int nef_theta_mlp_fwd(const float* theta, const float* W, const float* bias, float* y, int N, int O,
                      nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(theta && W && bias && y, NEF_E_NULL);
    NEF_REQUIRE(N > 0 && O > 0, NEF_E_SHAPE);
    hipLaunchKernelGGL(theta_mlp_fwd_kernel, dim3(nef_stream_grid((int64_t)N * O, 256)), dim3(256), 0, NEF_ST, theta, W,
                       bias, y, N, O);
    return nef_launch_status();
}

int nef_theta_mlp_bwd(const float* theta, const float* gy, float* gW, float* gb, int N, int O, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(theta && gy && gW && gb, NEF_E_NULL);
    NEF_REQUIRE(N > 0 && O > 0, NEF_E_SHAPE);
    hipLaunchKernelGGL(theta_mlp_bwd_kernel, dim3(O), dim3(256), 0, NEF_ST, theta, gy, gW, gb, N, O);
    return nef_launch_status();
}

}  // extern "C"
