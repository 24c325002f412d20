// conv_bww_glds.hip -- the Winograd weight gradients (transposed F(3,4) for K = 3, the 4 + 3 tap split F(4,4) / F(3,4) for
// K = 7) with their (gY, X) tiles streamed by LDS-DMA (`buffer_load_dwordx4 ... lds`) through a ring of LDS buffers.
//
// Why a second kernel next to conv_bwd_weight_kernel (conv_mfma.hip): these forms issue only 48..56 matrix instructions
// per wave and 64-column tile, so the ONE tile of register prefetch of that kernel is about as long as an HBM miss under
// load, and the timing-only builds put the exposed fetch + the staging stores at 25..27 % of the launch
// (profiles/r03_bwd_weight_ablation.md).  A deeper register ring does not fit next to 96..112 accumulator registers at two
// workgroups per CU.  LDS-DMA needs no staging registers and no ds_write pass, so the depth is bounded by LDS alone:
// NBUF buffers of one 32-column tile each keep NBUF - 1 tiles in flight per workgroup behind counted `s_waitcnt vmcnt`
// and raw `s_barrier`s (a __syncthreads() would drain the queue: its fence waits for vmcnt(0)).
//
// One workgroup = 64 output channels x 64 input channels of one group, 4 waves as 2 (co) x 2 (ci), each wave a 32 x 32
// tile per Winograd plane.  LDS image of a tile (the DMA writes lane-linear: 64 lanes x 16 bytes per instruction):
//     gY image [64 rows][9 chunks of 4 floats]   chunk c of row r = gy[r][t0 + 4c ..]      (chunk 8 is padding)
//     X  image [64 rows][XCH chunks]             chunk c of row r = x[r][t0 - PAD + 4c ..]  (K = 3: 9, K = 7: 10 + 1 padding)
// The odd chunk pitch makes the fragment reads (ds_read_b128: a lane's own row, 16 bytes) conflict-free in the hardware's
// 16-lane groups.  Tiles at the two ends of a sample fetch columns of the neighbouring rows; a patch pass (those tiles
// only, one extra barrier) overwrites them with zeros before the tile is used.  Chunks that are not entirely inside the
// tensor (first row / last row of the whole operand) are not fetched at all: the patch pass brings their valid elements
// in by lane-masked dword DMAs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "nefnet_hip.h"
#include "nef_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef NEF_GL_ABL
#define NEF_GL_ABL 0      // timing-only builds (results wrong): 1 = no DMA inside the loop, 2 = fragments read once per tile
#endif

#ifndef NEF_GL_OCC
#define NEF_GL_OCC 3      // workgroups per CU the register allocation aims at (168 VGPRs)
#endif

namespace {

#ifndef NEF_GL_TW
#define NEF_GL_TW 32
#endif
constexpr int TW = NEF_GL_TW;          // reduction columns per staged tile
constexpr int ROWS = 64;               // gY rows (co) and X rows (ci) of a workgroup
constexpr int GCH = (TW / 4) | 1;      // 16-byte chunks per gY image row (TW / 4 fetched + 1 so that the pitch is odd)

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// WINO: 2 = K 3, transposed F(3,4); 4 / 5 = the two launches of K 7 (taps 0..3 through the transposed F(4,4), taps 4..6
// through the transposed F(3,4)) -- the matrices of conv_bwd_weight_kernel.  AFF (K = 3): the BatchNorm affine + ReLU of
// the producing layer is applied to the X fragments (nef_conv_bwd_weight_wino's pro_mode 1).
// UP (K = 3): x is stored at half resolution [..][T/2] and upsampled x2 while the fragments are formed
// (nn.Upsample(scale_factor=2, mode='linear', align_corners=False): pro_mode 2) -- the X image then holds the T/2-resolution
// samples x[t0/2 - 1 .. t0/2 + 18), and a quad's six inputs are interpolated from four of them.
template <int K, bool AFF, int WINO, int NBUF, bool UP = false>
__global__ __launch_bounds__(256, NEF_GL_OCC) void conv_bww_glds_kernel(
    const float* __restrict__ x, int64_t x_bs, int64_t x_gs, const float* __restrict__ gy, int64_t gy_bs, int64_t gy_gs,
    float* __restrict__ ws, int B, int T, int G, int Cig, int Cog, int tps, int n_tiles, int m_tiles, int ci_chunks, int S,
    const float* __restrict__ pro_a, const float* __restrict__ pro_b, int pro_Bp, int n_pass, int64_t x_extent,
    int64_t gy_extent) {
    static_assert((WINO == 2 && K == 3) || ((WINO == 4 || WINO == 5) && K == 7 && !AFF), "forms");
    static_assert(!UP || K == 3, "the upsampling prologue: K = 3");
    constexpr int PAD = (K - 1) / 2;       // UP: the image starts one half-resolution sample before t0 / 2, also PAD = 1
    constexpr int XN = UP ? TW / 2 + 2 : TW + K - 1;       // positions of an X image row that are used
    constexpr int XUSE = (XN + 3) / 4;                      // chunks of an X row that are fetched
    const int Tin = UP ? (T >> 1) : T;                      // stored row length of x
    constexpr int XCH = XUSE | 1;                           // ... and its (odd) chunk pitch
    constexpr int GP = ROWS * GCH / 64, XP = ROWS * XCH / 64;      // DMA instructions ("pieces") per image
    constexpr int NP = GP + XP;
    constexpr int NPW = (NP + 3) / 4;                       // pieces per wave (the last one may not exist: NP % 4 waves have it)
    constexpr int GIMG = ROWS * GCH * 4, XIMG = ROWS * XCH * 4;    // floats
    constexpr int BUF = GIMG + XIMG;
    constexpr int NACC = WINO == 4 ? 7 : 6;
    static_assert(ROWS * GCH % 64 == 0 && ROWS * XCH % 64 == 0, "whole pieces");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tab = smem + NBUF * BUF;       // AFF: [n_pass][2][64] prologue constants of this workgroup's input channels

    // workgroup -> (unit = (split, group), member = (co tile, ci chunk)).  The members of a unit walk the SAME tiles -- co tiles
    // share the X tile, ci chunks the gY tile -- and workgroup ids go round-robin over the 8 XCDs (one L2 each): members are
    // placed 8 ids apart, i.e. on one XCD, dispatched together, so that all but the first find a tile in that L2.  (With
    // only the ci chunks co-located, as in conv_bwd_weight_kernel, the two co tiles of a 128-channel layer fetched every X
    // tile from HBM twice: 1.69 GB per launch of the K = 7 gradient for 0.98 GB of operands.)
    int bid = blockIdx.x;
    int member;
    {
        const int members = ci_chunks * m_tiles;
        const int units = (int)gridDim.x / members;
        const int full = (units / 8) * 8 * members;
        if (bid < full) {
            const int grp = bid / (8 * members), r = bid % (8 * members);
            member = r / 8;
            bid = grp * 8 + (r % 8);
        } else {
            const int r = bid - full;
            member = r % members;
            bid = (units / 8) * 8 + r / members;
        }
    }
    const int cc = member % ci_chunks, mt = member / ci_chunks;
    const int g = bid % G;
    const int split = bid / G;
    const int m0 = mt * ROWS, c0 = cc * ROWS;
    const int lane = threadIdx.x & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lo = lane & 31, hi = lane >> 5;
    const int wco = wave_u & 1, wci = wave_u >> 1;

    if constexpr (AFF) {
        for (int i = (int)threadIdx.x; i < n_pass * 128; i += 256) {
            const int p = i >> 7, r = i & 127;
            const float* src = (r & 64) ? pro_b : pro_a;
            tab[i] = src[(int64_t)p * G * Cig + (int64_t)g * Cig + c0 + (r & 63)];
        }
        __syncthreads();      // no DMA in flight yet
    }

    // per-lane byte offset of this lane's chunk inside the slab, per piece (pieces wave, wave + 4, ...)
    unsigned vo[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int p = wave_u + 4 * i;
        const bool isx = p >= GP;
        const int id = 64 * (isx ? p - GP : p) + lane;
        const int pitch = isx ? XCH : GCH;
        const int row = id / pitch, c = id - row * pitch;
        const bool dead = p >= NP || c >= (isx ? XUSE : TW / 4);
        vo[i] = dead ? NEF_OOB : (unsigned)((row * (isx ? Tin : T) + 4 * c) * 4);
    }

    f32x16 acc[NACC];
#pragma unroll
    for (int n = 0; n < NACC; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

    // tiles of this workgroup: a CONTIGUOUS range (first, first + 1, ...).  Rows of T = 1250 / 2500 / 5000 floats start 8..32
    // bytes off a 128-byte line, so the 128-byte row segment of a 32-column tile straddles two lines on most rows: with
    // neighbouring tiles on the same workgroup the shared line is fetched from HBM once and hit in L2 a tile later (a
    // strided assignment puts neighbours on different XCDs, i.e. L2s: 1.5..1.75x the algorithmic traffic).
    const int per = (n_tiles + S - 1) / S;
    const int first = split * per;
    const int n_k = first + per <= n_tiles ? per : (n_tiles > first ? n_tiles - first : 0);
    int ib0 = first / tps, itq = first - ib0 * tps;       // issue side: (sample, tile of the sample)
    int cb0 = ib0, ctq = itq;                             // compute side
#define NEF_GL_ISSUE(BUFI)                                                                                            \
    {                                                                                                               \
        const int t0_ = itq * TW, xt0_ = UP ? (t0_ >> 1) : t0_;                                                     \
        const int64_t xoff_ = (int64_t)ib0 * x_bs + (int64_t)g * x_gs + (int64_t)c0 * Tin - PAD;                    \
        const int64_t goff_ = (int64_t)ib0 * gy_bs + (int64_t)g * gy_gs + (int64_t)m0 * T;                          \
        const __amdgpu_buffer_rsrc_t xr_ = nef_rsrc(x + xoff_), gr_ = nef_rsrc(gy + goff_);                         \
        const bool edge_ = (itq == 0) || (xt0_ - PAD + 4 * XUSE > Tin) || (t0_ + TW > T);    /* a FETCHED image reaches outside its row */ \
        float* dst_ = smem + (BUFI) * BUF;                                                                          \
        _Pragma("unroll") for (int i = 0; i < NPW; ++i) {                                                           \
            const int p = wave_u + 4 * i;                                                                           \
            if (p < NP) {                                                                                           \
                unsigned v_ = vo[i];                                                                                \
                const bool isx = p >= GP;                                                                           \
                if (edge_ && v_ != NEF_OOB) {      /* a chunk that is not entirely inside the operand is not fetched */ \
                    const int64_t e0 = (isx ? xoff_ + xt0_ : goff_ + t0_) + (int64_t)(v_ >> 2);                     \
                    if (e0 < 0 || e0 + 3 >= (isx ? x_extent : gy_extent)) v_ = NEF_OOB;                             \
                }                                                                                                   \
                auto* l_ = (__attribute__((address_space(3))) void*)(dst_ + p * 256);                               \
                if (isx) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr_, l_, 16, (int)v_, xt0_ * 4, 0, 0);            \
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(gr_, l_, 16, (int)v_, t0_ * 4, 0, 0);                 \
            }                                                                                                       \
        }                                                                                                           \
        if (++itq == tps) {                                                                                         \
            itq = 0;                                                                                                \
            ++ib0;                                                                                                  \
        }                                                                                                           \
    }
#pragma unroll
    for (int k = 0; k < NBUF - 1; ++k)
        if (k < n_k) NEF_GL_ISSUE(k)

    const bool full_w = (NP % 4 == 0) || (wave_u < NP % 4);      // this wave issues NPW pieces per tile (else NPW - 1)
    int buf = 0;
    for (int k = 0; k < n_k; ++k) {
        // tile k has landed once at most the pieces of the tiles issued after it are outstanding
        {
            const int after = n_k - 1 - k;
            if (NEF_GL_ABL & 1) {
                wait_vm<0>();
            } else if (after >= NBUF - 2) {
                if (full_w) wait_vm<(NBUF - 2) * NPW>(); else wait_vm<(NBUF - 2) * (NPW - 1)>();
            } else if (NBUF > 3 && after == NBUF - 3) {
                if (full_w) wait_vm<(NBUF - 3) * NPW>(); else wait_vm<(NBUF - 3) * (NPW - 1)>();
            } else {
                wait_vm<0>();
            }
        }
        __builtin_amdgcn_s_barrier();       // every wave's pieces of tile k are in LDS; every wave is done with tile k - 1
        if (!(NEF_GL_ABL & 1) && k + NBUF - 1 < n_k) {
            const int nb = buf == 0 ? NBUF - 1 : buf - 1;          // the buffer tile k - 1 was read from
            NEF_GL_ISSUE(nb)
        }
        float* gimg = smem + buf * BUF;
        float* ximg = gimg + GIMG;
        const int t0 = ctq * TW, xt0 = UP ? (t0 >> 1) : t0;
        const bool edge_c = ctq == 0 || xt0 - PAD + 4 * XUSE > Tin || t0 + TW > T;      // same predicate as the issue side: the last
                                                         // tile, and the one before it when its halo (or the tail of its last chunk) crosses T
        if (edge_c) {
            // patch pass: columns outside [0, T) came from the neighbouring rows (or were not fetched): zero them -- a NaN
            // where the affine + ReLU prologue follows (max(NaN, 0) = 0: zero padding comes AFTER the prologue)
            const int row = (int)threadIdx.x >> 2, sub = (int)threadIdx.x & 3;
            if constexpr (!UP)      // (UP: the fragment code clamps and masks by index instead)
                for (int p = sub; p < XN; p += 4) {
                    const int t = t0 - PAD + p;
                    if (t < 0 || t >= T) ximg[row * (XCH * 4) + p] = AFF ? __builtin_nanf("") : 0.f;
                }
            for (int p = sub; p < TW; p += 4)
                if (t0 + p >= T) gimg[row * (GCH * 4) + p] = 0.f;
            // ... and the valid elements of the (at most three) chunks the issue side did not fetch because they reach outside
            // the operand -- the first chunk of its very first row, the chunk holding the end of its very last row --
            // come in by dword DMAs with only the lanes of those elements enabled (LDS address = base + 4 * lane)
            if (wave_u == 0) {
                bool any = false;
                if (PAD > 0 && ctq == 0 && cb0 == 0 && g == 0 && c0 == 0) {
                    if (lane < 4 - PAD)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(nef_rsrc(x), (__attribute__((address_space(3))) void*)(ximg + PAD), 4,
                                                                 lane * 4, 0, 0, 0);
                    any = true;
                }
                if (ctq != 0 && cb0 == B - 1 && g == G - 1) {
                    const int nx = Tin - xt0 + PAD, ng = T - t0;          // valid leading positions of an image row
                    if (c0 + ROWS == Cig && nx < 4 * XUSE && (nx & 3)) {
                        const float* src = x + (int64_t)cb0 * x_bs + (int64_t)g * x_gs + (int64_t)(Cig - 1) * Tin + xt0 - PAD + (nx & ~3);
                        if (lane < (nx & 3))
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                                nef_rsrc(src), (__attribute__((address_space(3))) void*)(ximg + (ROWS - 1) * (XCH * 4) + (nx & ~3)), 4,
                                lane * 4, 0, 0, 0);
                        any = true;
                    }
                    if (m0 + ROWS == Cog && ng < TW && (ng & 3)) {
                        const float* src = gy + (int64_t)cb0 * gy_bs + (int64_t)g * gy_gs + (int64_t)(Cog - 1) * T + t0 + (ng & ~3);
                        if (lane < (ng & 3))
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                                nef_rsrc(src), (__attribute__((address_space(3))) void*)(gimg + (ROWS - 1) * (GCH * 4) + (ng & ~3)), 4,
                                lane * 4, 0, 0, 0);
                        any = true;
                    }
                }
                if (any) wait_vm<0>();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        const float* ga = gimg + (wco * 32 + lo) * (GCH * 4) + 4 * hi;
        const float* xb = ximg + (wci * 32 + lo) * (XCH * 4) + 4 * hi;
        float pa = 1.f, pb = 0.f;
        if constexpr (AFF) {
            const int pass = cb0 / pro_Bp;
            pa = tab[pass * 128 + wci * 32 + lo];
            pb = tab[pass * 128 + 64 + wci * 32 + lo];
        }
        constexpr int NSTEP = (NEF_GL_ABL & 4) ? 0 : TW / 8;      // ABL 4: no fragment reads, transforms, MFMAs at all
        if constexpr (WINO == 4) {
            // taps 0..3: quad j = 2s + hi, gy[4j .. 4j+3] against d_m = x[4j-3+m], m = 0..6 (image positions 4j + m)
            f32x4 fg[2], fa[2], fb[2];
#define NEF_GL_LOAD(S_, BI)                                                                                          \
    {                                                                                                               \
        fg[BI] = *reinterpret_cast<const f32x4*>(ga + 8 * (S_));                                                    \
        fa[BI] = *reinterpret_cast<const f32x4*>(xb + 8 * (S_));                                                    \
        fb[BI] = *reinterpret_cast<const f32x4*>(xb + 8 * (S_) + 4);                                                \
    }
            if (NSTEP) NEF_GL_LOAD(0, 0)
#pragma unroll
            for (int s_ = 0; s_ < NSTEP; ++s_) {
                if (!(NEF_GL_ABL & 2) && s_ + 1 < NSTEP) NEF_GL_LOAD(s_ + 1, (s_ + 1) & 1)
                const float g0 = fg[s_ & 1][0], g1 = fg[s_ & 1][1], g2 = fg[s_ & 1][2], g3 = fg[s_ & 1][3];
                const float d0 = fa[s_ & 1][0], d1 = fa[s_ & 1][1], d2 = fa[s_ & 1][2], d3 = fa[s_ & 1][3];
                const float d4 = fb[s_ & 1][0], d5 = fb[s_ & 1][1], d6 = fb[s_ & 1][2];
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
                    u[6] = g3;
                }
                // B^T d of F(4,4) (points 0, 1, -1, 2, -2, 1/2, inf), exactly as conv_bwd_weight_kernel<.., 4> forms it
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
                for (int n = 0; n < 7; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[n], v[n], acc[n], 0, 0, 0);
            }
#undef NEF_GL_LOAD
        } else {
            // transposed F(3,4): gy[4j .. 4j+3] against x[4j-1 .. 4j+4] (K = 3: image positions 4j .. 4j+5) resp., for taps
            // 4..6 of K = 7, x[4j+1 .. 4j+6] (image positions 4j+4 .. 4j+9)
            constexpr int XO = WINO == 5 ? 4 : 0;
            f32x4 fg[2], fa[2];
            f32x2 fb[2];
            const float* xh = ximg + (wci * 32 + lo) * (XCH * 4) + 2 * hi;      // UP: quad j = 2s + hi reads image positions 2j .. 2j+3
#define NEF_GL_LOAD(S_, BI)                                                                                          \
    {                                                                                                               \
        fg[BI] = *reinterpret_cast<const f32x4*>(ga + 8 * (S_));                                                    \
        if constexpr (UP) {                                                                                         \
            const f32x2 h01_ = *reinterpret_cast<const f32x2*>(xh + 4 * (S_));                                      \
            fb[BI] = *reinterpret_cast<const f32x2*>(xh + 4 * (S_) + 2);                                            \
            fa[BI][0] = h01_[0];                                                                                    \
            fa[BI][1] = h01_[1];                                                                                    \
        } else {                                                                                                    \
            fa[BI] = *reinterpret_cast<const f32x4*>(xb + 8 * (S_) + XO);                                           \
            fb[BI] = *reinterpret_cast<const f32x2*>(xb + 8 * (S_) + XO + 4);                                       \
        }                                                                                                           \
    }
            if (NSTEP) NEF_GL_LOAD(0, 0)
#pragma unroll
            for (int s_ = 0; s_ < NSTEP; ++s_) {
                if (!(NEF_GL_ABL & 2) && s_ + 1 < NSTEP) NEF_GL_LOAD(s_ + 1, (s_ + 1) & 1)
                const float g0 = fg[s_ & 1][0], g1 = fg[s_ & 1][1], g2 = fg[s_ & 1][2], g3 = fg[s_ & 1][3];
                float d0 = fa[s_ & 1][0], d1 = fa[s_ & 1][1], d2 = fa[s_ & 1][2], d3 = fa[s_ & 1][3];
                float d4 = fb[s_ & 1][0], d5 = fb[s_ & 1][1];
                if constexpr (UP) {
                    // h0..h3 = x[m0 .. m0+3], m0 = tb/2 - 1 for the quad's first column tb; column t = tb - 1 + m is
                    // 0.75 * x[(t-1)/2] + 0.25 * x[(t+1)/2] for odd t, 0.25 * x[t/2 - 1] + 0.75 * x[t/2] for even t, indices clamped
                    // to [0, T/2) -- (1 - lambda) * x[i0] + lambda * x[i1] of conv_bwd_weight_kernel, same rounding
                    float h0 = d0, h1 = d1, h2 = d4, h3 = d5;
                    if constexpr (AFF) {      // the producing layer's affine + ReLU acts on the stored samples, before the interpolation
                        h0 = fmaxf(fmaf(h0, pa, pb), 0.f);
                        h1 = fmaxf(fmaf(h1, pa, pb), 0.f);
                        h2 = fmaxf(fmaf(h2, pa, pb), 0.f);
                        h3 = fmaxf(fmaf(h3, pa, pb), 0.f);
                    }
                    const int tb = t0 + 4 * (2 * s_ + hi);
                    if (edge_c) {
                        const int m0 = (tb >> 1) - 1;
                        if (m0 < 0) h0 = h1;
                        if (m0 + 1 > Tin - 1) h1 = h0;
                        if (m0 + 2 > Tin - 1) h2 = h1;
                        if (m0 + 3 > Tin - 1) h3 = h2;
                    }
                    d0 = 0.75f * h0 + 0.25f * h1;
                    d1 = 0.25f * h0 + 0.75f * h1;
                    d2 = 0.75f * h1 + 0.25f * h2;
                    d3 = 0.25f * h1 + 0.75f * h2;
                    d4 = 0.75f * h2 + 0.25f * h3;
                    d5 = 0.25f * h2 + 0.75f * h3;
                    if (edge_c) {      // the two clamped columns are the sample itself (lambda = 0), columns outside [0, T) are the conv's zero padding
                        const float w75[6] = {h0, h1, h1, h2, h2, h3};
                        float* dm[6] = {&d0, &d1, &d2, &d3, &d4, &d5};
#pragma unroll
                        for (int m = 0; m < 6; ++m) {
                            const int t = tb - 1 + m;
                            if (t == 0 || t == T - 1) *dm[m] = w75[m];
                            if (t < 0 || t >= T) *dm[m] = 0.f;
                        }
                    }
                }
                if constexpr (AFF && !UP) {
                    d0 = fmaxf(fmaf(d0, pa, pb), 0.f);
                    d1 = fmaxf(fmaf(d1, pa, pb), 0.f);
                    d2 = fmaxf(fmaf(d2, pa, pb), 0.f);
                    d3 = fmaxf(fmaf(d3, pa, pb), 0.f);
                    d4 = fmaxf(fmaf(d4, pa, pb), 0.f);
                    d5 = fmaxf(fmaf(d5, pa, pb), 0.f);
                }
                float u[6], v[6];
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
                const float t1 = fmaf(-4.f, d2, d4), t2 = fmaf(-4.f, d1, d3);
                const float t3 = d4 - d2, t4 = 2.f * (d3 - d1);
                v[0] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
                v[1] = t1 + t2;
                v[2] = t1 - t2;
                v[3] = t3 + t4;
                v[4] = t3 - t4;
                v[5] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
#pragma unroll
                for (int n = 0; n < 6; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[n], v[n], acc[n], 0, 0, 0);
            }
#undef NEF_GL_LOAD
        }
        if (++ctq == tps) {
            ctq = 0;
            ++cb0;
        }
        buf = buf + 1 == NBUF ? 0 : buf + 1;
    }
#undef NEF_GL_ISSUE

    // partials: ws[split][g][k][co][ci], gW = G^T M as in conv_bwd_weight_kernel
    const int ci = c0 + wci * 32 + lo;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if constexpr (WINO == 4 || WINO == 5) if ((k < 4) != (WINO == 4)) continue;      // a launch owns its tap group only
        float* dst = ws + ((((int64_t)split * G + g) * K + k) * Cog) * Cig;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = m0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const float m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r], m4 = acc[4][r], m5 = acc[5][r];
            float v;
            if constexpr (WINO == 4) {
                v = (k == 0) ? fmaf(-0.5f, acc[0][r], fmaf(m5, 32.f / 45.f, fmaf(m3, 1.f / 36.f, fmaf(m4, -1.f / 60.f, (m2 * (1.f / 9.f) - m1 * (1.f / 3.f))))))
                  : (k == 1) ? fmaf(m5, 16.f / 45.f, fmaf(m3, 1.f / 18.f, fmaf(m4, 1.f / 30.f, -(m2 * (1.f / 9.f) + m1 * (1.f / 3.f)))))
                  : (k == 2) ? fmaf(m5, 8.f / 45.f, fmaf(m3, 1.f / 9.f, fmaf(m4, -1.f / 15.f, (m2 * (1.f / 9.f) - m1 * (1.f / 3.f)))))
                             : fmaf(m5, 4.f / 45.f, fmaf(m3, 2.f / 9.f, fmaf(m4, 2.f / 15.f, -(m2 * (1.f / 9.f) + m1 * (1.f / 3.f))))) + acc[NACC - 1][r];
            } else {
                const int kk = WINO == 5 ? k - 4 : k;
                const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                v = (kk == 0) ? fmaf(0.25f, acc[0][r], fmaf(s34, 1.f / 24.f, -s12 * (1.f / 6.f)))
                  : (kk == 1) ? fmaf(d34, 1.f / 12.f, -d12 * (1.f / 6.f))
                              : (s34 - s12) * (1.f / 6.f) + m5;
            }
            dst[(int64_t)co * Cig + ci] = v;
        }
    }
}

template <int K, bool AFF, int WINO, int NBUF, bool UP = false>
int launch(const float* x, int64_t x_bs, int64_t x_gs, const float* gy, int64_t gy_bs, int64_t gy_gs, float* ws, int B, int T,
           int G, int Cig, int Cog, const float* pro_a, const float* pro_b, int pro_Bp, int n_pass, int S_max, int fixed_S,
           int* S_used, hipStream_t st) {
    constexpr int XCH = (((UP ? TW / 2 + 2 : TW + K - 1) + 3) / 4) | 1;
    constexpr size_t lds_tiles = (size_t)NBUF * (ROWS * GCH + ROWS * XCH) * 16;
    const size_t lds = lds_tiles + (AFF ? (size_t)n_pass * 128 * sizeof(float) : 0);
    const void* fn = reinterpret_cast<const void*>(&conv_bww_glds_kernel<K, AFF, WINO, NBUF, UP>);
    static unsigned long long lds_set = 0;
    if (int e = nef_ensure_dyn_lds(fn, lds_tiles + 8 * 128 * sizeof(float), &lds_set)) return e;
    const int m_tiles = Cog / ROWS, ci_chunks = Cig / ROWS;
    const int tps = (T + TW - 1) / TW;
    const int n_tiles = B * tps;
    int S = fixed_S;
    if (S <= 0) {      // exactly one round of resident workgroups, never more splits than the workspace was sized for
        static int resident_dev[64] = {0};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        int resident = __atomic_load_n(&resident_dev[dev & 63], __ATOMIC_ACQUIRE);
        if (resident == 0) {
            int per_cu = 0;
            // cached per device: priced with the LARGEST request this instantiation can make (8 passes of prologue table), so
            // the cached figure holds for every later call whatever its n_pass
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds_tiles + 8 * 128 * sizeof(float)) != hipSuccess ||
                per_cu <= 0)
                per_cu = 2;
            resident = per_cu * nef_cu_count();
            __atomic_store_n(&resident_dev[dev & 63], resident, __ATOMIC_RELEASE);
        }
        S = resident / (G * m_tiles * ci_chunks);
        if (S > S_max) S = S_max;
        if (S > n_tiles) S = n_tiles;
        if (S < 1) S = 1;
    }
    *S_used = S;
    const int64_t x_extent = (int64_t)(B - 1) * x_bs + (int64_t)(G - 1) * x_gs + (int64_t)Cig * (UP ? T / 2 : T);
    const int64_t gy_extent = (int64_t)(B - 1) * gy_bs + (int64_t)(G - 1) * gy_gs + (int64_t)Cog * T;
    const int64_t blocks = (int64_t)S * G * m_tiles * ci_chunks;
    hipLaunchKernelGGL((conv_bww_glds_kernel<K, AFF, WINO, NBUF, UP>), dim3((unsigned)blocks), dim3(256), lds, st, x, x_bs, x_gs, gy,
                       gy_bs, gy_gs, ws, B, T, G, Cig, Cog, tps, n_tiles, m_tiles, ci_chunks, S, pro_a, pro_b, pro_Bp, n_pass,
                       x_extent, gy_extent);
    return nef_launch_status();
}

}  // namespace

// Ring depth.  Measured (tools/bench_conv.py, all eight weight-gradient shapes of the step): 2, 3 and 4 buffers, 32- and
// 64-column tiles, two and three workgroups per CU all land within 3 % of each other -- with the tiles' HBM traffic at the
// algorithmic bytes the fetch is no longer a latency problem.  Two buffers at three workgroups per CU is the smallest.
#ifndef NEF_GLDS_NBUF
#define NEF_GLDS_NBUF 2
#endif

// Shapes the LDS-DMA kernel takes (everything else stays on conv_bwd_weight_kernel): whole 64-channel slabs, at least two
// tiles per sample, no in_scale, T % 4 == 0 with the upsampling prologue; at most 8 BatchNorm passes in the prologue table.
extern "C" __attribute__((visibility("hidden"))) bool nef_bww_glds_ok(int B, int T, int Cig, int Cog, int K, int pro_mode, int pro_Bp,
                                                                      bool in_scale) {
    static const int on = [] {
        const char* e = nef_diag_env("NEF_BWW_GLDS");
        return (e && e[0] == '0') ? 0 : 1;
    }();
    if (!on || in_scale || (K != 3 && K != 7) || (K == 7 && pro_mode != 0)) return false;
    if ((pro_mode & 2) && T % 4 != 0) return false;
    if (Cig % ROWS != 0 || Cog % ROWS != 0 || T < 2 * TW || T % 2 != 0) return false;
    if ((int64_t)(ROWS - 1) * T * 4 + 4 * 11 * 4 >= 0x7FFFFFFCll) return false;      // per-lane offsets are 32-bit
    if ((pro_mode & 1) && (pro_Bp <= 0 || (B + pro_Bp - 1) / pro_Bp > 8)) return false;
    return true;
}

// `half`: 0 for K = 3; 4 / 5 for the two launches of K = 7.  S_max: the split count the workspace was sized for; fixed_S > 0:
// use exactly this many (second launch of K = 7).  Partials go to ws[S][G][K][Cog][Cig] like conv_bwd_weight_kernel's.
extern "C" __attribute__((visibility("hidden"))) int nef_bww_glds_launch(
    const float* x, int64_t x_bs, int64_t x_gs, const float* gy, int64_t gy_bs, int64_t gy_gs, float* ws, int B, int T, int G,
    int Cig, int Cog, int K, int half, const float* pro_a, const float* pro_b, int pro_mode, int pro_Bp, int S_max, int fixed_S,
    int* S_used, hipStream_t st) {
    const int n_pass = (pro_mode & 1) ? (B + pro_Bp - 1) / pro_Bp : 0;
    if (K == 3) {
        if (pro_mode == 3)
            return launch<3, true, 2, NEF_GLDS_NBUF, true>(x, x_bs, x_gs, gy, gy_bs, gy_gs, ws, B, T, G, Cig, Cog, pro_a, pro_b, pro_Bp,
                                                           n_pass, S_max, fixed_S, S_used, st);
        if (pro_mode & 2)
            return launch<3, false, 2, NEF_GLDS_NBUF, true>(x, x_bs, x_gs, gy, gy_bs, gy_gs, ws, B, T, G, Cig, Cog, nullptr, nullptr, 1,
                                                            0, S_max, fixed_S, S_used, st);
        if (pro_mode & 1)
            return launch<3, true, 2, NEF_GLDS_NBUF>(x, x_bs, x_gs, gy, gy_bs, gy_gs, ws, B, T, G, Cig, Cog, pro_a, pro_b, pro_Bp,
                                                     n_pass, S_max, fixed_S, S_used, st);
        return launch<3, false, 2, NEF_GLDS_NBUF>(x, x_bs, x_gs, gy, gy_bs, gy_gs, ws, B, T, G, Cig, Cog, nullptr, nullptr, 1, 0,
                                                  S_max, fixed_S, S_used, st);
    }
    if (half == 4)
        return launch<7, false, 4, NEF_GLDS_NBUF>(x, x_bs, x_gs, gy, gy_bs, gy_gs, ws, B, T, G, Cig, Cog, nullptr, nullptr, 1, 0,
                                                  S_max, fixed_S, S_used, st);
    return launch<7, false, 5, NEF_GLDS_NBUF>(x, x_bs, x_gs, gy, gy_bs, gy_gs, ws, B, T, G, Cig, Cog, nullptr, nullptr, 1, 0, S_max,
                                              fixed_S, S_used, st);
}
