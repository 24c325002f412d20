// Heartbeat-segment ROI ops on the latent time axis.
//   roi_align : reference codes/network/utils/roi_pooling_1d.py:38-69 (`roi_algin`).  The reference feeds
//               F.grid_sample a [B,C,T,1] image and a grid whose x component carries the ROI positions, so x
//               addresses the size-1 axis and y = 0 addresses the middle of time.  This kernel
//               evaluates exactly that bilinear sample with zero padding, align_corners=False.
//   roi_unpool: roi_pooling_1d.py:72-99 (`roi_pooling_reverse`): per (sample, segment) linear resampling
//               32 -> len_j (align_corners=False) concatenated along time.  Integer bookkeeping (:82-92) is
//               (long)(float(roi) * 0.25f), bit-exact with the reference.
// rois stay int64 on the device; no host loop, no per-segment launches.
#include "nef_common.h"

namespace {

constexpr int NSEG = NEF_N_SEG;
constexpr int BINS = NEF_ROI_BINS;
constexpr int SEGW = 2 * NEF_ROI_BINS;   // 32 samples per decoded segment

__device__ __forceinline__ int64_t latent_index(int64_t roi) {
    return (int64_t)((float)roi * 0.25f);   // rois.float().mul_(0.25).long()  (:82-85)
}

// grid x for (segment j, bin s): torch.linspace(r0, r1, 16) on fp32, r = roi*0.25*(2/T) - 1   (:50-58)
__device__ __forceinline__ float grid_x(const int64_t* __restrict__ roi_b, int j, int s, int T) {
    const float sc = (float)(2.0 / (double)T);
    const float r0 = ((float)roi_b[2 * j] * 0.25f) * sc + (-1.0f);
    const float r1 = ((float)roi_b[2 * j + 1] * 0.25f) * sc + (-1.0f);
    const float step = (r1 - r0) / (float)(BINS - 1);
    return s < BINS / 2 ? r0 + step * (float)s : r1 - step * (float)(BINS - 1 - s);
}

// bilinear weight of the single column (W == 1) for normalised x, zero padding
__device__ __forceinline__ float col_weight(float gx) {
    const float ix = ((gx + 1.f) * 1.f - 1.f) * 0.5f;
    const float x0 = floorf(ix);
    const float we = ix - x0;          // weight of column x0+1
    const float ww = 1.f - we;         // weight of column x0
    float w = 0.f;
    if (x0 == 0.f) w += ww;
    if (x0 + 1.f == 0.f) w += we;
    return w;
}

struct RowTap { int r0, r1; float w0, w1; };

__device__ __forceinline__ RowTap row_taps(int T) {
    const float iy = ((0.f + 1.f) * (float)T - 1.f) * 0.5f;
    const float y0 = floorf(iy);
    RowTap rt;
    rt.r0 = (int)y0;
    rt.r1 = rt.r0 + 1;
    rt.w1 = iy - y0;
    rt.w0 = 1.f - rt.w1;
    if (rt.r0 < 0 || rt.r0 >= T) rt.w0 = 0.f;
    if (rt.r1 < 0 || rt.r1 >= T) { rt.w1 = 0.f; rt.r1 = rt.r0; }
    if (rt.r0 < 0 || rt.r0 >= T) rt.r0 = rt.r1;
    return rt;
}

// one thread per output element; out [B][C][7][16]
// z may hold only a window of the time axis: zT stored samples per row starting at time t_off (the two rows read
// are the only ones roi_algin ever touches, so the producer can skip the rest -- see engine.py).
__global__ void roi_align_fwd_kernel(const float* __restrict__ z, const int64_t* __restrict__ rois,
                                     float* __restrict__ out, int B, int C, int T, int zT, int t_off) {
    const int64_t n = (int64_t)B * C * NSEG * BINS;
    const RowTap rt = row_taps(T);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(i % BINS);
        const int j = (int)((i / BINS) % NSEG);
        const int64_t bc = i / (BINS * NSEG);
        const int b = (int)(bc / C);
        const float wx = col_weight(grid_x(rois + (int64_t)b * NSEG * 2, j, s, T));
        const float* zr = z + bc * zT - t_off;
        // grid_sample accumulates nw*(n*w) + ne*.. + sw*(s*w) + se*..; with W == 1 only one column is in range
        out[i] = zr[rt.r0] * (rt.w0 * wx) + zr[rt.r1] * (rt.w1 * wx);
    }
}

// gz [B][C][T]: zero except the two middle rows
__global__ void roi_align_bwd_kernel(const float* __restrict__ gout, const int64_t* __restrict__ rois,
                                     float* __restrict__ gz, int B, int C, int T, int zT, int t_off) {
    const int64_t rows = (int64_t)B * C;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const RowTap rt = row_taps(T);
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const int b = (int)(row / C);
        float acc = 0.f;
        for (int e = lane; e < NSEG * BINS; e += 64) {
            const int j = e / BINS, s = e % BINS;
            const float wx = col_weight(grid_x(rois + (int64_t)b * NSEG * 2, j, s, T));
            acc = fmaf(gout[row * NSEG * BINS + e], wx, acc);
        }
        acc = nef_wave_sum(acc);
        float* gr = gz + row * zT - t_off;
        for (int t = t_off + lane; t < t_off + zT; t += 64) {
            float v = 0.f;
            if (t == rt.r0) v += acc * rt.w0;
            if (t == rt.r1 && rt.w1 != 0.f) v += acc * rt.w1;
            gr[t] = v;
        }
    }
}

struct SegTable { int start[NSEG]; int len[NSEG]; int off[NSEG]; float scale[NSEG]; };

__device__ __forceinline__ bool load_segments(const int64_t* __restrict__ roi_b, int T, SegTable& st) {
    int run = 0;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NSEG; ++j) {
        const int64_t a = latent_index(roi_b[2 * j]);
        const int64_t e = latent_index(roi_b[2 * j + 1]);
        int64_t len = e - a;
        if (len < 0) { ok = false; len = 0; }
        if (run + len > T) { ok = false; len = T - run; }
        st.start[j] = (int)a;
        st.len[j] = (int)len;
        st.off[j] = run;
        st.scale[j] = len > 0 ? (float)SEGW / (float)(int)len : 0.f;      // F.interpolate's input/output ratio
        run += (int)len;
    }
    if (run != T) ok = false;
    return ok;
}

// F.interpolate(mode='linear', align_corners=False) source index for output i of a len-long segment
__device__ __forceinline__ void lerp_src(int i, float scale, int& i0, int& i1, float& l0, float& l1) {
    float src = scale * ((float)i + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    if (i0 > SEGW - 1) i0 = SEGW - 1;
    i1 = i0 + (i0 < SEGW - 1 ? 1 : 0);
    l1 = src - (float)i0;
    if (l1 < 0.f) l1 = 0.f;
    if (l1 > 1.f) l1 = 1.f;
    l0 = 1.f - l1;
}

// one wave per (b, c) row of the output; zseg [B][C][7][32] -> out [B][C][T].
// The pass is VALU-bound, not HBM-bound (PMC: 1.7e8 vector instructions for 1.2e8 outputs in the first version, which
// searched the segment of every output with a 7-way select chain): the wave therefore walks the row SEGMENT BY SEGMENT --
// the segment's offset, length and scale are wave-uniform (scalar registers), a lane only evaluates lerp_src for its
// own output and gathers two of the segment's 32 samples.
__global__ void roi_unpool_fwd_kernel(const float* __restrict__ zseg, const int64_t* __restrict__ rois,
                                      float* __restrict__ out, int32_t* __restrict__ status, int B, int C, int T) {
    const int64_t rows = (int64_t)B * C;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    // a wave takes a contiguous run of rows: they mostly belong to one sample, whose segment table (about 70 vector
    // instructions per segment: i64 <-> f32 conversions and a division) is then derived once, not once per row
    const int64_t per_wave = (rows + (int64_t)gridDim.x * 4 - 1) / ((int64_t)gridDim.x * 4);
    const int64_t row_lo = ((int64_t)blockIdx.x * 4 + wave) * per_wave;
    const int64_t row_hi = row_lo + per_wave < rows ? row_lo + per_wave : rows;
    int b_cur = -1;
    SegTable st;
    for (int64_t row = row_lo; row < row_hi; ++row) {
        const int b = (int)(row / C);
        if (b != b_cur) {
            b_cur = b;
            const bool ok = load_segments(rois + (int64_t)b * NSEG * 2, T, st);
            if (!ok && status && lane == 0) status[0] = 1;
        }
        const float* zr = zseg + row * NSEG * SEGW;
        float* orow = out + row * T;
        // the row's 7 x 32 samples live in four registers of the wave (sample e in register e / 64, lane e % 64); the two
        // taps of an output come over the cross-lane network (ds_bpermute) instead of two dependent memory gathers
        float r[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = (q * 64 + lane < NSEG * SEGW) ? zr[q * 64 + lane] : 0.f;
        int covered = 0;
#pragma unroll
        for (int j = 0; j < NSEG; ++j) {
            const int len = st.len[j], off = st.off[j];
            const float sc = st.scale[j];
            const float reg = r[j >> 1];
            const int base = (j & 1) * SEGW;
            for (int ib = 0; ib < len; ib += 64) {                 // wave-uniform trip count: every lane feeds the shuffles
                const int i = ib + lane;
                int i0, i1;
                float l0, l1;
                lerp_src(i, sc, i0, i1, l0, l1);
                const float v0 = __shfl(reg, base + i0), v1 = __shfl(reg, base + i1);
                if (i < len) orow[off + i] = l0 * v0 + l1 * v1;
            }
            covered = off + len;
        }
        for (int t = covered + lane; t < T; t += 64) orow[t] = 0.f;     // segments that do not reach T (flagged in status)
    }
}

// first output i in [0, len] of a segment whose source index i0(i) reaches s (i0 is non-decreasing in i): the closed form
// of lerp_src's `scale*(i+0.5)-0.5 >= s`, then corrected by evaluating lerp_src itself around the guess, so that the
// transpose partitions the outputs exactly as the forward assigned them whatever the rounding of the closed form.
// lerp_src's left tap alone (its clamp to 31 does not matter against s <= 31)
__device__ __forceinline__ int unpool_left_tap(int i, float scale) {
    float src = scale * ((float)i + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    return (int)src;
}

__device__ __forceinline__ int unpool_first_reaching(int s, int len, float scale, float inv_scale) {
    if (s <= 0) return 0;
    int i = (int)ceilf(((float)s + 0.5f) * inv_scale - 0.5f);
    if (i < 0) i = 0;
    if (i > len) i = len;
    // the closed form and the forward's rounded expression can disagree by one position when the boundary falls within
    // rounding of an integer: one verified step either way ...
    if (i > 0 && unpool_left_tap(i - 1, scale) >= s) --i;
    else if (i < len && unpool_left_tap(i, scale) < s) ++i;
    // ... and, should that ever not be enough, the plain search (never taken in practice; keeps the partition exact)
    if ((i > 0 && unpool_left_tap(i - 1, scale) >= s) || (i < len && unpool_left_tap(i, scale) < s)) {
#pragma nounroll
        while (i > 0 && unpool_left_tap(i - 1, scale) >= s) --i;
#pragma nounroll
        while (i < len && unpool_left_tap(i, scale) < s) ++i;
    }
    return i;
}

// gather form of the transpose: lane s of segment j collects the outputs that read sample s:
//   gz[j][s] = sum_{i: i0(i)=s} l0(i) g[i] + sum_{i: i1(i)=s} l1(i) g[i],  {i0 = s} = [first(s), first(s+1)),
//   {i1 = s} = {i0 = s-1} (plus {i0 = 31} for s = 31, where i1 is clamped) -- two short contiguous ranges instead of a
//   widened candidate window with a test per candidate (VALU-bound before: 985 vector instructions per element).
// STAGED = true: the wave first streams its gradient row into a private LDS strip (coalesced, all loads in flight at
// once), then gathers from LDS -- the strided global gathers of the direct form are latency-bound (PMC: 60 % of wave
// cycles parked).  STAGED = false reads the row in place and serves rows too long for the strip (T > 4096).
template <bool STAGED>
__global__ __launch_bounds__(256) void roi_unpool_bwd_kernel(const float* __restrict__ gout,
                                                             const int64_t* __restrict__ rois,
                                                             float* __restrict__ gzseg, int B, int C, int T) {
    extern __shared__ float strip_lds[];
    const int64_t rows = (int64_t)B * C;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int s = lane & (SEGW - 1), half = lane >> 5;          // two segments per trip: lanes 0..31 and 32..63
    const int64_t per_wave = (rows + (int64_t)gridDim.x * 4 - 1) / ((int64_t)gridDim.x * 4);     // see the forward
    const int64_t row_lo = ((int64_t)blockIdx.x * 4 + wave) * per_wave;
    const int64_t row_hi = row_lo + per_wave < rows ? row_lo + per_wave : rows;
    int b_cur = -1;
    SegTable st;
    for (int64_t row = row_lo; row < row_hi; ++row) {
        const int b = (int)(row / C);
        if (b != b_cur) {
            b_cur = b;
            load_segments(rois + (int64_t)b * NSEG * 2, T, st);
        }
        const float* grow = gout + row * T;
        const float* gr = grow;
        if constexpr (STAGED) {
            float* strip = strip_lds + wave * T;
            // NU loads per lane in flight at once: a T = 1250 row (20 x 64 floats) is one trip -- with 5 per trip the wave
            // paid four HBM round trips per row (33 % of the HBM roof, round 2)
            constexpr int NU = 20;
            for (int t0 = 0; t0 < T; t0 += 64 * NU) {
                float v[NU];
#pragma unroll
                for (int u = 0; u < NU; ++u) v[u] = (t0 + u * 64 + lane < T) ? grow[t0 + u * 64 + lane] : 0.f;
#pragma unroll
                for (int u = 0; u < NU; ++u)
                    if (t0 + u * 64 + lane < T) strip[t0 + u * 64 + lane] = v[u];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            gr = strip;
        }
        float* gz = gzseg + row * NSEG * SEGW;
#pragma unroll
        for (int jj = 0; jj < NSEG + 1; jj += 2) {
            const int j = jj + half;
            if (j >= NSEG) continue;
            const int len = half ? st.len[jj + 1 < NSEG ? jj + 1 : jj] : st.len[jj];
            const int off = half ? st.off[jj + 1 < NSEG ? jj + 1 : jj] : st.off[jj];
            const float sc = half ? st.scale[jj + 1 < NSEG ? jj + 1 : jj] : st.scale[jj];
            float acc = 0.f;
            if (len > 0) {
                const int b0 = unpool_first_reaching(s, len, sc, (float)len * (1.f / SEGW));
                const int nb = __shfl_down(b0, 1), pb = __shfl_up(b0, 1);   // neighbours lie in the same 32-lane half
                const int b1 = s + 1 < SEGW ? nb : len;
                const int a0 = s > 0 ? pb : b0;
                // outputs in [a0, b0) read s as their RIGHT tap: weight l1 = src - (s - 1), in [0, 1) without a clamp because
                // their left tap is s - 1; outputs in [b0, b1) read it as their LEFT tap: weight l0 = 1 - (src - s) -- and at
                // s = 31 the right tap is clamped onto the left one, the two weights add up to 1.  src is lerp_src's
                // expression term for term ((float)i + 0.5f is exact, so the running fi below is too).
                const float* gp = gr + off;
                float fi = (float)a0 + 0.5f;
                const float sm1 = (float)(s - 1);
                for (int i = a0; i < b0; ++i, fi += 1.f) {
                    float src = sc * fi - 0.5f;
                    src = src < 0.f ? 0.f : src;
                    acc += (src - sm1) * gp[i];
                }
                const float c0 = s == SEGW - 1 ? 1.f : (float)(s + 1), k = s == SEGW - 1 ? 0.f : 1.f;
                for (int i = b0; i < b1; ++i, fi += 1.f) {
                    float src = sc * fi - 0.5f;
                    src = src < 0.f ? 0.f : src;
                    acc += (c0 - k * src) * gp[i];
                }
            }
            gz[j * SEGW + s] = acc;
        }
        if constexpr (STAGED) __builtin_amdgcn_wave_barrier();     // the strip is rewritten by the next row
    }
}

__global__ void roi_segment_table_kernel(const int64_t* __restrict__ rois, int64_t* __restrict__ seg_start,
                                         int64_t* __restrict__ seg_len, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t a = latent_index(rois[2 * i]);
    const int64_t e = latent_index(rois[2 * i + 1]);
    seg_start[i] = a;
    seg_len[i] = e - a;
}

}  // namespace

#define NEF_ST ((hipStream_t)stream)

extern "C" {

static bool window_covers_taps(int T, int zT, int t_off) {
    const int r0 = (T - 1) / 2, r1 = (r0 + 1 < T) ? r0 + 1 : r0;
    return zT > 0 && t_off >= 0 && t_off <= r0 && r1 < t_off + zT && t_off + zT <= T;
}

int nef_roi_align_fwd(const float* z, const int64_t* rois, float* out, int B, int C, int T, int zT, int t_off,
                      nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(z && rois && out, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && C > 0 && T > 0 && window_covers_taps(T, zT, t_off), NEF_E_SHAPE);
    const int64_t n = (int64_t)B * C * NSEG * BINS;
    hipLaunchKernelGGL(roi_align_fwd_kernel, dim3(nef_stream_grid(n, 256)), dim3(256), 0, NEF_ST, z, rois, out, B, C, T,
                       zT, t_off);
    return nef_launch_status();
}

int nef_roi_align_bwd(const float* gout, const int64_t* rois, float* gz, int B, int C, int T, int zT, int t_off,
                      nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(gout && rois && gz, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && C > 0 && T > 0 && window_covers_taps(T, zT, t_off), NEF_E_SHAPE);
    hipLaunchKernelGGL(roi_align_bwd_kernel, dim3(nef_stream_grid((int64_t)B * C, 4)), dim3(256), 0, NEF_ST, gout, rois,
                       gz, B, C, T, zT, t_off);
    return nef_launch_status();
}

int nef_roi_unpool_fwd(const float* zseg, const int64_t* rois, float* out, int32_t* status, int B, int C, int T,
                       nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(zseg && rois && out, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && C > 0 && T > 0, NEF_E_SHAPE);
    hipLaunchKernelGGL(roi_unpool_fwd_kernel, dim3(nef_stream_grid((int64_t)B * C, 4)), dim3(256), 0, NEF_ST, zseg,
                       rois, out, status, B, C, T);
    return nef_launch_status();
}

int nef_roi_unpool_bwd(const float* gout, const int64_t* rois, float* gzseg, int B, int C, int T,
                       nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(gout && rois && gzseg, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && C > 0 && T > 0, NEF_E_SHAPE);
    const dim3 grid(nef_stream_grid((int64_t)B * C, 4));
    if (T <= 4096)
        hipLaunchKernelGGL(roi_unpool_bwd_kernel<true>, grid, dim3(256), (size_t)4 * T * sizeof(float), NEF_ST, gout,
                           rois, gzseg, B, C, T);
    else
        hipLaunchKernelGGL(roi_unpool_bwd_kernel<false>, grid, dim3(256), 0, NEF_ST, gout, rois, gzseg, B, C, T);
    return nef_launch_status();
}

int nef_roi_segment_table(const int64_t* rois, int64_t* seg_start, int64_t* seg_len, int B, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(rois && seg_start && seg_len, NEF_E_NULL);
    NEF_REQUIRE(B > 0, NEF_E_SHAPE);
    const int n = B * NSEG;
    hipLaunchKernelGGL(roi_segment_table_kernel, dim3((n + 255) / 256), dim3(256), 0, NEF_ST, rois, seg_start, seg_len,
                       n);
    return nef_launch_status();
}

}  // extern "C"
