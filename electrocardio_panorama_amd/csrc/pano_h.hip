// Half-precision panorama decoder (SURVEY §8 f2 / BASELINE configs 4 and 5): the eval-mode view sweep
// `Model_nefnet.forward(phase='test')` / `gen_ecg` (reference codes/network/model_nefnet.py:181-190, :196-218) decodes
// one latent at Q query angles; in eval mode BatchNorm is a fixed affine that folds into the conv, so one decoder pass is
//   up2 -> conv(256->128)+ReLU -> conv(128->128)+ReLU -> up2 -> conv(128->64)+ReLU -> conv(64->64)+ReLU -> conv(64->1)
//   -> sigmoid(x/3)                                                                  (model_nefnet.py:101-107, :168)
// Here the four wide convs run on the gfx950 fp16 matrix cores (v_mfma_f32_32x32x16_f16, fp32 accumulate) with
// activations stored TIME-MAJOR in half precision, `[pair][time][channel]`: 8 consecutive channels of one time step are
// one 16-byte vector, which is exactly one lane's MFMA B operand, and a x2 linear upsample along time is a per-row
// blend.  The reference has no reduced-precision behaviour; this path is gated against the fp32 path at 2e-3 rel-L2
// (tests/test_pano_gpu.py) and is opt-in (Model_nefnet.panorama_dtype).
//
// hconv_kernel<CIN,COUT,PRO>: implicit GEMM  Y[co][t] = sum_{tap,ci} W[co][ci][tap] * X[t+tap-1][ci]
//   block   = 256 threads (4 waves), persistent: walks tiles blockIdx.x, +gridDim.x, ... (grid = resident blocks)
//   tile    = all COUT rows x NT time columns of ONE pair (NT = 128 for COUT=128, 256 for COUT=64)
//   wave    = 64 co x 64 t  -> 2 x 2 accumulators of v_mfma_f32_32x32x16_f16 (64 accumulator registers)
//   K loop  = CIN/64 channel chunks x 3 taps; per (chunk, tap) "stage" 4 MFMA k-steps of 16 channels
//   LDS     = X chunk [(NT+2) rows][64 ch + 8 pad] halfs (staged once per chunk, all 3 taps read it shifted by a row;
//             the 16-byte pad makes the 16-row ds_read_b128 groups conflict-free)
//             + the chunk's 3 x COUT x 64 weight halfs in MFMA-fragment order (1 KB per fragment, lane-linear)
//   pipeline: the X rows and the weights of the NEXT (tile, chunk) travel global -> registers while the current
//             chunk's 48 MFMAs per wave run, two barriers per chunk (buffer loads whose descriptor spans exactly one
//             pair: the conv zero padding is the hardware's out-of-range zero; weights are L2-resident, shared by
//             every block, and stay in LDS for the whole kernel when the layer has a single chunk)
//   PRO bit0: multiply channel ci by scale[pair][ci] while staging (the per-angle query scaling, model_nefnet.py:184-186)
//   PRO bit1: X is the x2 linear upsample (align_corners=False) of the stored rows: raw rows go to an LDS scratch and
//             are blended from there with packed-half fma
//   epilogue: + bias, ReLU, -> fp16, transposed through LDS (over the X chunk, 64 channels per pass) so that every
//             output row leaves as 16-byte vectors
//   OUT=1   : the 64->64 layer also evaluates the last conv (64 -> 1) on its staged tile (see below)
// Round 2: layers 1-3 (Cin >= 128) run on hconv_wide_kernel below; hconv_kernel keeps the 64 -> 64 layer with the fused
// last conv.  Measured (MI355X, BASELINE config 4: 1024 samples x 360 angles, len 512): 53.9 ms per sweep against 377 ms
// on the fp32 path; per 16384 pairs L1 736 us (1.11 PFLOP/s), L2 472 us, L3 543 us, L4 + last conv 324 us
// (profiles/r02_pano_fp16_kernel_stats.md).
#include "nef_common.h"

typedef _Float16 nef_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 nef_h4 __attribute__((ext_vector_type(4)));
typedef float nef_f16acc __attribute__((ext_vector_type(16)));
typedef float nef_f8 __attribute__((ext_vector_type(8)));

#define PH_XRS 144   // bytes per staged X row: 64 halfs + 16 B pad (conflict-free ds_read_b128 across 16 rows)

// ------------------------------------------------------------------------------------------------------------
// fp32 [B][C][T] -> fp16 [B][T][C]  (the latent enters the half-precision pipeline once per sweep)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ph_transpose_kernel(const float* __restrict__ x, _Float16* __restrict__ y, int C,
                                                           int T) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z, c0 = blockIdx.y * 64, t0 = blockIdx.x * 64;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const float* xb = x + (size_t)b * C * T;
    for (int r = ly; r < 64; r += 4) {
        const int c = c0 + r, t = t0 + lx;
        tile[r][lx] = (c < C && t < T) ? xb[(size_t)c * T + t] : 0.f;
    }
    __syncthreads();
    _Float16* yb = y + (size_t)b * T * C;
    for (int r = ly; r < 64; r += 4) {
        const int t = t0 + r, c = c0 + lx;
        if (t < T && c < C) yb[(size_t)t * C + c] = (_Float16)tile[lx][r];
    }
}

// ------------------------------------------------------------------------------------------------------------
// fp32 weights [Cout][Cin][3] -> fp16 MFMA A fragments, stage-major:
//   wp[(((cc*3 + tap)*4 + kq)*MT + mt)*64 + lane][e] = w[mt*32 + (lane&31)][cc*64 + kq*16 + 8*(lane>>5) + e][tap]
// ------------------------------------------------------------------------------------------------------------
__global__ void ph_pack_weight_kernel(const float* __restrict__ w, _Float16* __restrict__ wp, int Cout, int Cin) {
    const int MT = Cout / 32;
    const int64_t total = (int64_t)Cout * Cin * 3;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int e = i & 7;
        int64_t f = i >> 3;
        const int lane = f & 63;
        f >>= 6;
        const int mt = f % MT;
        f /= MT;
        const int kq = f & 3;
        f >>= 2;
        const int tap = f % 3;
        const int cc = f / 3;
        const int co = mt * 32 + (lane & 31);
        const int ci = cc * 64 + kq * 16 + 8 * (lane >> 5) + e;
        wp[i] = (_Float16)w[((size_t)co * Cin + ci) * 3 + tap];
    }
}

// Persistent version: a block walks tiles blockIdx.x, +gridDim.x, ...; the X rows of the NEXT (tile, channel chunk) are
// in flight (global -> registers) while the matrix cores work on the current one, and the weight stage after the
// current one likewise, so that neither HBM nor L2 latency sits on a block's critical path.
// OUT = 1 (64 -> 64 layer only): the tile never leaves the chip -- the last conv (64 -> 1, k3) is evaluated on the staged
// fp16 tile and its pre-activation goes to `logit` (fp32, addressed like the final output): interior columns by plain
// stores, the two edge columns of a tile by atomic adds, because their neighbour tap lives in the adjacent tile, which
// adds its share the same way (exactly two addends on a zeroed word: order-independent, hence deterministic).
template <int CIN, int COUT, int PRO, int OUT, int NI, int MINB>
__global__ __launch_bounds__(256, MINB) void hconv_kernel(const _Float16* __restrict__ x, const nef_h8* __restrict__ wp,
                                                       const float* __restrict__ bias, const float* __restrict__ scale,
                                                       _Float16* __restrict__ y, int T, int tiles_per_n, int total_tiles,
                                                       int x_div, int nq, long sc_bs, long sc_is,
                                                       const float* __restrict__ wout, float* __restrict__ logit,
                                                       long out_bs, long out_is) {
    static_assert(!OUT || (COUT == 64 && PRO == 0), "the fused last conv rides on the 64-channel layer");
    constexpr int WM = COUT / 64;            // waves along the output-channel axis
    constexpr int WN = 4 / WM;               // waves along time
    constexpr int NT = WN * NI * 32;         // time columns per tile (128 for COUT=128, 256 for COUT=64)
    constexpr int MT = COUT / 32;            // 32-row A fragments per k-step
    constexpr int XROWS = NT + 2;
    constexpr int XBYTES = XROWS * PH_XRS;
    constexpr int SROWS = (PRO & 2) ? NT / 2 + 4 : XROWS;   // rows fetched from memory per (tile, chunk)
    constexpr int SBYTES = (PRO & 2) ? SROWS * PH_XRS : 0;  // raw source rows, blended into Xl
    constexpr int XIT = (SROWS * 8 + 255) / 256;            // fetched 16-byte items per thread
    constexpr int WST_V = COUT * 64 / 8;     // h8 vectors per weight stage
    constexpr int WPT = WST_V / 256;         // ... per thread
    constexpr int NCC = CIN / 64;
    constexpr int NPASS = COUT / 64;         // epilogue passes: 64 output channels per staged row
    static_assert(XBYTES % 16 == 0 && SBYTES % 16 == 0, "LDS regions must stay 16-byte aligned");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Xl = smem;                         // [XROWS][PH_XRS]; the epilogue's output staging aliases it
    char* Sl = smem + XBYTES;                // [SROWS][PH_XRS] (upsampling prologue only)
    nef_h8* Wl = (nef_h8*)(smem + XBYTES + SBYTES);   // [3 taps][WST_V]: the weights of one channel chunk
    float* Ol = (float*)(smem + XBYTES + SBYTES + 3 * WST_V * 16);   // OUT: wout[192], d0[NT], d2[NT]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int Tin = (PRO & 2) ? T / 2 : T;
    const int seg = tid & 7;                 // this thread's 8-channel segment of a staged row (constant: 256 % 8 == 0)
    const int brow = wn * NI * 32 + (lane & 31);
    const int bcol = 16 * (lane >> 5);       // byte offset of this lane's 8 k-values inside a 16-channel k-step

    nef_f16acc acc[2][NI];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    nef_h8 xr[XIT];
    float qr[8];
    nef_h8 hzero;
#pragma unroll
    for (int e = 0; e < 8; ++e) hzero[e] = (_Float16)0.f;

    // fetch (tile, chunk) into registers: rows at SOURCE resolution.  One buffer descriptor per pair whose extent is
    // exactly that pair's rows: a row before the sequence start (negative -> huge unsigned offset) or past its end is
    // out of range for the hardware and reads as zero -- the conv zero padding costs no compare and no address register.
    const unsigned xvoff = (unsigned)((tid >> 3) * CIN * 2 + seg * 16);
#define PH_FETCH(tile_, cc_)                                                                                  \
    {                                                                                                         \
        const int n_ = (tile_) / tiles_per_n, t0_ = ((tile_) % tiles_per_n) * NT;                             \
        const __amdgpu_buffer_rsrc_t xd_ = __builtin_amdgcn_make_buffer_rsrc(                                 \
            const_cast<_Float16*>(x + (size_t)(n_ / x_div) * Tin * CIN), 0, Tin * CIN * 2, 0x00020000);       \
        const int u_ = ((PRO & 2) ? t0_ / 2 - 2 : t0_ - 1) * CIN * 2 + (cc_) * 128;                           \
        _Pragma("unroll") for (int j = 0; j < XIT; ++j) {                                                     \
            unsigned o_ = xvoff + (unsigned)(u_ + j * 32 * CIN * 2);                                          \
            if (j == XIT - 1 && tid + j * 256 >= SROWS * 8) o_ = NEF_OOB;                                     \
            xr[j] = __builtin_bit_cast(nef_h8, __builtin_amdgcn_raw_buffer_load_b128(xd_, (int)o_, 0, 0));    \
        }                                                                                                     \
        if (PRO & 1) {                                                                                        \
            const float* sc_ = scale + (size_t)(n_ / nq) * sc_bs + (size_t)(n_ % nq) * sc_is + (cc_) * 64;    \
            const __amdgpu_buffer_rsrc_t sd_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sc_), 0, 256, 0x00020000); \
            const nef_f32x4 q0_ = nef_buf_f32x4(sd_, seg * 32, 0), q1_ = nef_buf_f32x4(sd_, seg * 32 + 16, 0); \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) { qr[e] = q0_[e]; qr[4 + e] = q1_[e]; }             \
        }                                                                                                     \
    }
    const __amdgpu_buffer_rsrc_t wd = nef_rsrc(wp);
#define PH_WFETCH(cc_)                                                                                        \
    _Pragma("unroll") for (int tp = 0; tp < 3; ++tp) _Pragma("unroll") for (int j = 0; j < WPT; ++j)          \
        wreg[tp][j] = __builtin_bit_cast(nef_h8, __builtin_amdgcn_raw_buffer_load_b128(                       \
            wd, tid * 16, ((cc_) * 3 + tp) * (WST_V * 16) + j * 4096, 0));

    nef_h8 wreg[3][WPT];
    PH_WFETCH(0);
    if (OUT && tid < 192) Ol[tid] = wout[tid];   // [ci][tap]; visible after the first barrier
    int tile = blockIdx.x;
    if (tile < total_tiles) PH_FETCH(tile, 0);
    bool wfresh = true;                      // wreg holds weights that are not in LDS yet

#pragma unroll 1
    for (; tile < total_tiles; tile += gridDim.x) {
        const int n = tile / tiles_per_n, t0 = (tile % tiles_per_n) * NT;
#pragma unroll 1
        for (int cc = 0; cc < NCC; ++cc) {
            __syncthreads();                 // every wave is done reading Xl / Wl (previous chunk, or the epilogue staging)
            if (wfresh) {
#pragma unroll
                for (int tp = 0; tp < 3; ++tp)
#pragma unroll
                    for (int j = 0; j < WPT; ++j) Wl[tp * WST_V + tid + j * 256] = wreg[tp][j];
                wfresh = NCC > 1;            // a single chunk's weights stay resident for the whole kernel
            }
            if (PRO & 2) {
#pragma unroll
                for (int j = 0; j < XIT; ++j) {
                    const int idx = tid + j * 256;
                    if (j < XIT - 1 || idx < SROWS * 8) *(nef_h8*)(Sl + (idx >> 3) * PH_XRS + seg * 16) = xr[j];
                }
                __syncthreads();
                // Upsample(scale 2, linear, align_corners=False): out[2i] = .25 x[i-1] + .75 x[i],
                // out[2i+1] = .75 x[i] + .25 x[i+1], indices clamped to the sequence; then the query scaling.
                // Staged row r is output time t = t0-1+r (t0 even): source row i = t>>1 sits at Sl row ((r-1)>>1)+2, its
                // partner one Sl row up (t odd <=> r even) or down; the clamp only bites at t = 0 and t = T-1, where
                // the partner is the row itself.  Packed-half arithmetic: .25*b is exact, the fma rounds once.
                nef_h8 qh;
                if (PRO & 1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) qh[e] = (_Float16)qr[e];
                }
                const int r_first = tid >> 3;
                const int dj = (r_first & 1) ? -PH_XRS : PH_XRS;
                constexpr int BIT = (XROWS * 8 + 255) / 256;
#pragma unroll
                for (int it = 0; it < BIT; ++it) {
                    const int r = r_first + it * 32, t = t0 - 1 + r;
                    if (it == BIT - 1 && r >= XROWS) continue;
                    nef_h8 v = hzero;
                    if (t >= 0 && t < T) {
                        const char* pa = Sl + (((r - 1) >> 1) + 2) * PH_XRS + seg * 16;
                        const nef_h8 a = *(const nef_h8*)pa;
                        const nef_h8 b = *(const nef_h8*)(pa + ((t == 0 || t == T - 1) ? 0 : dj));
                        nef_h8 c75;
#pragma unroll
                        for (int e = 0; e < 8; ++e) c75[e] = (_Float16)0.75f;
                        v = __builtin_elementwise_fma(a, c75, b * (_Float16)0.25f);
                        if (PRO & 1) v = v * qh;
                    }
                    *(nef_h8*)(Xl + r * PH_XRS + seg * 16) = v;
                }
            } else {
#pragma unroll
                for (int j = 0; j < XIT; ++j) {
                    const int idx = tid + j * 256;
                    nef_h8 v = xr[j];
                    if (PRO & 1) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (_Float16)((float)v[e] * qr[e]);
                    }
                    if (j < XIT - 1 || idx < SROWS * 8) *(nef_h8*)(Xl + (idx >> 3) * PH_XRS + seg * 16) = v;
                }
            }
            // Next chunk's operands, in flight during this chunk's 48 MFMAs (and the epilogue).  Weights FIRST: loads
            // retire in order, and the L2-resident weights must not queue behind rows that come from HBM.
            if (NCC > 1) PH_WFETCH(cc + 1 < NCC ? cc + 1 : 0);
            if (cc + 1 < NCC) {
                PH_FETCH(tile, cc + 1);
            } else if (tile + (int)gridDim.x < total_tiles) {
                PH_FETCH(tile + (int)gridDim.x, 0);
            }
            __syncthreads();                 // X chunk and the chunk's weights are visible
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                const nef_h8* Ws = Wl + tap * WST_V;
                const char* Xs = Xl + (brow + tap) * PH_XRS + bcol;
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) {
                    nef_h8 a[2], b[NI];
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) a[mi] = Ws[(kq * MT + wm * 2 + mi) * 64 + lane];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) b[ni] = *(const nef_h8*)(Xs + ni * 32 * PH_XRS + kq * 32);
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
                }
            }
        }

        // epilogue: bias + ReLU -> fp16, staged [t][64 channels] in LDS (over Xl), then whole 16-byte vectors out
        _Float16* yb = y + (size_t)n * T * COUT;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            __syncthreads();                 // Xl free: all MFMA reads (or the previous pass's stores) are done
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                if (NPASS == 2 && mi != p) continue;
                const int slot0 = (NPASS == 2) ? wm * 32 : mi * 32;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cw = 8 * g + 4 * (lane >> 5);
                    const nef_f32x4 bv = *(const nef_f32x4*)(bias + wm * 64 + mi * 32 + cw);
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        nef_h4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            o[e] = (_Float16)fmaxf(acc[mi][ni][g * 4 + e] + bv[e], 0.f);
                            acc[mi][ni][g * 4 + e] = 0.f;
                        }
                        *(nef_h4*)(Xl + (wn * NI * 32 + ni * 32 + (lane & 31)) * PH_XRS + (slot0 + cw) * 2) = o;
                    }
                }
            }
            __syncthreads();
            if (OUT) {
                // last conv on the staged tile: row r = this thread; d_k = sum_c wout[c][k] * tile[r][c]
                const int r = tid, t = t0 + r;
                float d0 = 0.f, d1 = 0.f, d2 = 0.f;
                if (r < NT && t < T) {
#pragma unroll
                    for (int sg = 0; sg < 8; ++sg) {
                        const nef_h8 v = *(const nef_h8*)(Xl + r * PH_XRS + sg * 16);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float f = (float)v[e];
                            const float* wc = Ol + (sg * 8 + e) * 3;
                            d0 = fmaf(wc[0], f, d0);
                            d1 = fmaf(wc[1], f, d1);
                            d2 = fmaf(wc[2], f, d2);
                        }
                    }
                }
                if (r < NT) {
                    Ol[192 + r] = d0;        // tap 0 weights this row into column t+1
                    Ol[192 + NT + r] = d2;   // tap 2 into column t-1
                }
                __syncthreads();
                float* lg = logit + (size_t)(n / nq) * out_bs + (size_t)(n % nq) * out_is;
                if (r < NT && t < T) {
                    const float s_ = d1 + (r > 0 ? Ol[192 + r - 1] : 0.f) + (r < NT - 1 ? Ol[192 + NT + r + 1] : 0.f);
                    if (r == 0 || r == NT - 1) atomicAdd(lg + t, s_);
                    else lg[t] = s_;
                    if (r == NT - 1 && t + 1 < T) atomicAdd(lg + t + 1, d0);
                    if (r == 0 && t > 0) atomicAdd(lg + t - 1, d2);
                }
            } else {
                for (int idx = tid; idx < NT * 8; idx += 256) {
                    const int r = idx >> 3;
                    const int slot = seg * 8;
                    const int co = (NPASS == 2) ? (slot >> 5) * 64 + p * 32 + (slot & 31) : slot;
                    if (t0 + r < T)
                        *(nef_h8*)(yb + (size_t)(t0 + r) * COUT + co) = *(const nef_h8*)(Xl + r * PH_XRS + seg * 16);
                }
            }
        }
    }
#undef PH_FETCH
#undef PH_WFETCH
}

// ------------------------------------------------------------------------------------------------------------
// hconv_wide_kernel<CIN,PRO>: the two 128-output-channel layers (256 -> 128 behind the x2 upsampling and the query
// scaling, 128 -> 128).  hconv_kernel spends more LDS cycles than matrix-core cycles on these (gfx950: ds_write_b128
// moves ~79 B/clk/CU, so re-staging the 48 KB weight chunk of every 128-column tile alone costs 620 of the tile's
// 1536 matrix cycles; reads run at 256 B/clk).  Here
//   wave    = 64 co x 128 t (2 x 4 accumulator tiles, 128 accumulator registers): half the A bytes per MFMA
//   tile    = 128 co x 256 t of one pair; 4 waves = 2 (co) x 2 (t); persistent blocks, 2 per CU
//   A       = weight fragments straight from L2 into registers (lane-linear 1 KB fragments, 2 per k-step, ring of three
//             k-steps in flight): the weights never touch LDS
//   B       = X chunk [258 rows][64 ch + pad] DOUBLE-buffered in LDS: the next (tile, chunk) is fetched into registers
//             at the top of a chunk and stored into the other buffer behind its 96 MFMAs -> one barrier per chunk
//   x2 upsample: a thread blends 8 (9 at the tile edges) output rows from 6 source rows in registers -- no raw-row
//             LDS round trip; same packed-half arithmetic as hconv_kernel (0.25*b exact, one fma rounding)
//   epilogue: per time half (128 rows x 128 co) through the consumed X buffer, whole 256-byte rows out
//   COUT = 64 (128 -> 64 behind the second upsampling): 4 waves along time, 64 co x 64 t each, one epilogue pass
// ------------------------------------------------------------------------------------------------------------
#define PHW_ORS 272   // bytes per staged output row: 128 halfs + 16 B pad

//   OUT = 1 (64 -> 64, round 5): the last conv (64 -> 1) evaluated on the staged output tile as in hconv_kernel -- the 64-channel
//             tile never reaches memory; tile-edge columns are completed by the neighbouring tile (two-addend atomic add)
template <int CIN, int COUT, int PRO, int OUT = 0>
__global__ __launch_bounds__(256, 2) void hconv_wide_kernel(const _Float16* __restrict__ x, const nef_h8* __restrict__ wp,
                                                         const float* __restrict__ bias, const float* __restrict__ scale,
                                                         _Float16* __restrict__ y, int T, int tiles_per_n, int total_tiles,
                                                         int x_div, int nq, long sc_bs, long sc_is,
                                                         const float* __restrict__ wout, float* __restrict__ logit,
                                                         long out_bs, long out_is) {
    static_assert(!OUT || (COUT == 64 && PRO == 0), "the fused last conv rides on the 64-channel layer");
    constexpr int WM = COUT / 64, WN = 4 / WM;   // waves along the output channels / along time
    constexpr int NT = 256, NI = NT / (32 * WN); // wave = 64 co x 128 t (COUT = 128) or 64 co x 64 t (COUT = 64)
    constexpr int MT = COUT / 32;                // A fragments per k-step in the packed weights
    constexpr int XROWS = NT + 2;
    constexpr int XBYTES = XROWS * PH_XRS;
    constexpr int NCC = CIN / 64;
    constexpr bool UP = (PRO & 2) != 0, SC = (PRO & 1) != 0;
    constexpr int XIT = UP ? 6 : 9;          // 16-byte fetches per thread and chunk ...
    constexpr int XB1 = UP ? 6 : 5;          // ... in two batches (0..XB1, XB1..XIT) when not upsampling: fewer registers
    constexpr int AD = 4;                    // A ring: fragments of AD - 1 k-steps in flight
    static_assert(COUT == 64 || COUT == 128, "one or two waves along the output channels");
    static_assert(128 * PHW_ORS <= XBYTES, "the output staging of one time half lives in one X buffer");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // two X buffers + a 16-byte dump slot per thread

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int lo = lane & 31, hi = lane >> 5;
    const int seg = tid & 7, rg = tid >> 3;
    const int Tin = UP ? T / 2 : T;
    char* const dump = smem + 2 * XBYTES + tid * 16;
    float* const Of = (float*)(smem + 2 * XBYTES + 256 * 16);      // OUT: wout[192], d0[NT], d2[NT]
    if (OUT && tid < 192) Of[tid] = wout[tid];                     // [ci][tap]; visible after the first barrier

    nef_f16acc acc[2][NI];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    nef_h8 xr[XB1];
    float qr[8];
    int ft0 = 0;                             // t0 of the tile held in xr
    nef_h8 hzero;
#pragma unroll
    for (int e = 0; e < 8; ++e) hzero[e] = (_Float16)0.f;

#define PHW_FETCH(tile_, cc_, J0_, J1_)                                                                       \
    {                                                                                                         \
        const int n_ = (tile_) / tiles_per_n, t0_ = ((tile_) % tiles_per_n) * NT;                             \
        ft0 = t0_;                                                                                            \
        const __amdgpu_buffer_rsrc_t xd_ = __builtin_amdgcn_make_buffer_rsrc(                                 \
            const_cast<_Float16*>(x + (size_t)(n_ / x_div) * Tin * CIN), 0, Tin * CIN * 2, 0x00020000);       \
        if constexpr (UP) {                                                                                   \
            const int rb_ = t0_ / 2 + 4 * rg - 1;                                                             \
            _Pragma("unroll") for (int k = 0; k < XIT; ++k) {                                                 \
                int r_ = rb_ + k;                                                                             \
                r_ = r_ < 0 ? 0 : (r_ > Tin - 1 ? Tin - 1 : r_);                                              \
                xr[k] = __builtin_bit_cast(nef_h8, __builtin_amdgcn_raw_buffer_load_b128(                     \
                    xd_, r_ * (CIN * 2) + seg * 16, (cc_) * 128, 0));                                         \
            }                                                                                                 \
        } else {                                                                                              \
            /* the whole offset in the per-lane operand: only that one takes part in the hardware range check */ \
            const unsigned o_ = (unsigned)((t0_ - 1 + rg) * (CIN * 2) + seg * 16 + (cc_) * 128);              \
            _Pragma("unroll") for (int j = (J0_); j < (J1_); ++j)                                             \
                xr[j - (J0_)] = __builtin_bit_cast(nef_h8, __builtin_amdgcn_raw_buffer_load_b128(             \
                    xd_, (int)((j == XIT - 1 && tid + j * 256 >= XROWS * 8) ? NEF_OOB : o_ + j * 32 * CIN * 2), 0, 0)); \
        }                                                                                                     \
        if (SC && (J0_) == 0) {                                                                               \
            const float* sc_ = scale + (size_t)(n_ / nq) * sc_bs + (size_t)(n_ % nq) * sc_is + (cc_) * 64;    \
            const __amdgpu_buffer_rsrc_t sd_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sc_), 0, 256, 0x00020000); \
            const nef_f32x4 q0_ = nef_buf_f32x4(sd_, seg * 32, 0), q1_ = nef_buf_f32x4(sd_, seg * 32 + 16, 0); \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) { qr[e] = q0_[e]; qr[4 + e] = q1_[e]; }             \
        }                                                                                                     \
    }
    // registers -> X buffer `Xn_` (row r of the buffer is output time ft0 - 1 + r), pieces J0_ <= j < J1_.  Upsampling:
    // j = 0..7 are the thread's eight blended rows, j = 8 the tile's two halo rows (threads of the first / last row
    // group); otherwise j counts the fetched vectors and JB_ is the first one of the batch held in xr.
#define PHW_STAGE(Xn_, J0_, J1_, JB_)                                                                         \
    {                                                                                                         \
        if constexpr (UP) {                                                                                   \
            nef_h8 qh, c75;                                                                                   \
            _Pragma("unroll") for (int e = 0; e < 8; ++e) { c75[e] = (_Float16)0.75f; qh[e] = SC ? (_Float16)qr[e] : (_Float16)1.f; } \
            _Pragma("unroll") for (int j = (J0_); j < (J1_) && j < 8; ++j) {                                  \
                const nef_h8 a_ = xr[(j >> 1) + 1], b_ = (j & 1) ? xr[(j >> 1) + 2] : xr[j >> 1];             \
                nef_h8 v_ = __builtin_elementwise_fma(a_, c75, b_ * (_Float16)0.25f);                         \
                if (SC) v_ = v_ * qh;                                                                         \
                if (ft0 + 8 * rg + j >= T) v_ = hzero;                                                        \
                *(nef_h8*)((Xn_) + (8 * rg + 1 + j) * PH_XRS + seg * 16) = v_;                                \
            }                                                                                                 \
            if ((J1_) > 8) {   /* t = ft0 - 1 (odd) and t = ft0 + 256 (even); the other row groups store into */ \
                const nef_h8 a_ = rg == 0 ? xr[0] : xr[5], b_ = rg == 0 ? xr[1] : xr[4];   /* their dump slot: */ \
                nef_h8 v_ = __builtin_elementwise_fma(a_, c75, b_ * (_Float16)0.25f);   /* no branch among the MFMAs */ \
                if (SC) v_ = v_ * qh;                                                                         \
                if (rg == 0 ? ft0 == 0 : ft0 + NT >= T) v_ = hzero;                                           \
                char* d_ = (Xn_) + (rg == 0 ? 0 : XROWS - 1) * PH_XRS + seg * 16;                             \
                *(nef_h8*)((rg == 0 || rg == 31) ? d_ : dump) = v_;                                           \
            }                                                                                                 \
        } else {                                                                                              \
            _Pragma("unroll") for (int j = (J0_); j < (J1_); ++j) {                                           \
                nef_h8 v_ = xr[j - (JB_)];                                                                    \
                if (SC) {                                                                                     \
                    _Pragma("unroll") for (int e = 0; e < 8; ++e) v_[e] = (_Float16)((float)v_[e] * qr[e]);   \
                }                                                                                             \
                char* d_ = (Xn_) + (rg + 32 * j) * PH_XRS + seg * 16;                                         \
                *(nef_h8*)((j < XIT - 1 || tid + j * 256 < XROWS * 8) ? d_ : dump) = v_;                      \
            }                                                                                                 \
        }                                                                                                     \
    }
    // A fragments of k-step s_ (= tap * 4 + kq) of chunk ccv_: stage-major packing, 4 fragments of 1 KB per stage
    const __amdgpu_buffer_rsrc_t wd = nef_rsrc(wp);
    const int avoff = lane * 16 + wm * 2048;
    nef_h8 a[AD][2];
#define PHW_A(slot_, ccv_, s_)                                                                                \
    _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                                          \
        a[slot_][mi] = __builtin_bit_cast(nef_h8, __builtin_amdgcn_raw_buffer_load_b128(                      \
            wd, avoff, ((ccv_) * 12 + (s_)) * (MT * 1024) + mi * 1024, 0));

    int tile = blockIdx.x;
    int p = 0;
    PHW_A(0, 0, 0)
    PHW_A(1, 0, 1)
    PHW_A(2, 0, 2)
    if (tile < total_tiles) {
        PHW_FETCH(tile, 0, 0, XB1)
        PHW_STAGE(smem, 0, UP ? 9 : XB1, 0)
        if (XB1 < XIT) {
            PHW_FETCH(tile, 0, XB1, XIT)
            PHW_STAGE(smem, XB1, XIT, XB1)
        }
    }
    __syncthreads();

#pragma unroll 1
    for (; tile < total_tiles; tile += gridDim.x) {
        const int n = tile / tiles_per_n, t0 = (tile % tiles_per_n) * NT;
#pragma unroll 1
        for (int cc = 0; cc < NCC; ++cc) {
            const char* Xc = smem + p * XBYTES;
            char* Xn = smem + (p ^ 1) * XBYTES;
            const bool more = cc + 1 < NCC || tile + (int)gridDim.x < total_tiles;
            const int ccn = cc + 1 < NCC ? cc + 1 : 0;
            // the (tile, chunk) staged behind this one; the block's very last chunk re-stages its own tile (unread) so
            // that the k-steps carry no branch
            const int tile_n = (cc + 1 < NCC || !more) ? tile : tile + (int)gridDim.x;
            const char* Xs = Xc + (wn * (NI * 32) + lo) * PH_XRS + 16 * hi;
            nef_h8 b[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[ni] = *(const nef_h8*)(Xs + ni * 32 * PH_XRS);
#pragma unroll
            for (int s = 0; s < 12; ++s) {
                // the A fragments AD - 1 k-steps ahead (wrapping into the next chunk: the weights are the same for every
                // tile).  Loads return in order, so a wait for A also waits for every X fetch issued before it: the X
                // batches go out right BEHIND an A issue and get AD - 1 k-steps before a wait reaches them.
                if (s + AD - 1 < 12) {
                    PHW_A((s + AD - 1) % AD, cc, s + AD - 1)
                } else {
                    PHW_A((s + AD - 1) % AD, ccn, s + AD - 1 - 12)
                }
                if (s == 1) PHW_FETCH(tile_n, ccn, 0, XB1)
                if (!UP && s == 7) PHW_FETCH(tile_n, ccn, XB1, XIT)
                __builtin_amdgcn_sched_barrier(0);   // keep the fetches where they are
                // the staging of the fetched rows rides in the shadow of the MFMAs, a few rows per k-step: by k-step 5
                // (10) a wait for A has already covered the first (second) batch
                if constexpr (UP) {
                    if (s >= 6 && s <= 9) PHW_STAGE(Xn, 2 * (s - 6), 2 * (s - 6) + 2, 0)
                    if (s == 10) PHW_STAGE(Xn, 8, 9, 0)
                } else {
                    if (s == 5) PHW_STAGE(Xn, 0, 2, 0)
                    if (s == 6) PHW_STAGE(Xn, 2, XB1, 0)
                    if (s == 10) PHW_STAGE(Xn, XB1, XB1 + 2, XB1)
                    if (s == 11) PHW_STAGE(Xn, XB1 + 2, XIT, XB1)
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s % AD][mi], b[ni], acc[mi][ni], 0, 0, 0);
                    if (s + 1 < 12)       // this B register is free again: next k-step's fragment
                        b[ni] = *(const nef_h8*)(Xs + (ni * 32 + (s + 1) / 4) * PH_XRS + ((s + 1) % 4) * 32);
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {    // issue order: 2 MFMAs, the LDS read that refills their B register,
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // a share of the staging arithmetic / stores, ...
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 40 / NI, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 4 / NI, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();                 // next chunk staged; every wave is done reading Xc
            p ^= 1;
        }

        // epilogue: bias + ReLU -> fp16 through the buffer the last chunk consumed; whole rows leave as 16-byte vectors
        char* Ol = smem + (p ^ 1) * XBYTES;
        _Float16* yb = y + (size_t)n * T * COUT;
        constexpr int NPS = COUT / 64;           // 128 channels: one time half (128 rows x 256 B) per pass
        constexpr int ORS = COUT == 128 ? PHW_ORS : PH_XRS;
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
            if (NPS == 1 || wn == ps) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int co = wm * 64 + mi * 32 + 8 * g + 4 * hi;
                        const nef_f32x4 bv = *(const nef_f32x4*)(bias + co);
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            nef_h4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                o[e] = (_Float16)fmaxf(acc[mi][ni][g * 4 + e] + bv[e], 0.f);
                                acc[mi][ni][g * 4 + e] = 0.f;
                            }
                            *(nef_h4*)(Ol + ((NPS == 1 ? wn * NI * 32 : 0) + ni * 32 + lo) * ORS + co * 2) = o;
                        }
                    }
            }
            __syncthreads();
            constexpr int VPR = COUT / 8;        // 16-byte vectors per output row
            constexpr int ROWS = NT / NPS;
            if constexpr (OUT) {
                // last conv on the staged tile: row r = this thread; d_k = sum_c wout[c][k] * tile[r][c]
                const int r = tid, t = t0 + r;
                float d0 = 0.f, d1 = 0.f, d2 = 0.f;
                if (t < T) {
#pragma unroll
                    for (int sg = 0; sg < 8; ++sg) {
                        const nef_h8 v = *(const nef_h8*)(Ol + r * ORS + sg * 16);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float f = (float)v[e];
                            const float* wc = Of + (sg * 8 + e) * 3;
                            d0 = fmaf(wc[0], f, d0);
                            d1 = fmaf(wc[1], f, d1);
                            d2 = fmaf(wc[2], f, d2);
                        }
                    }
                }
                Of[192 + r] = d0;        // tap 0 weights this row into column t + 1
                Of[192 + NT + r] = d2;   // tap 2 into column t - 1
                __syncthreads();
                float* lg = logit + (size_t)(n / nq) * out_bs + (size_t)(n % nq) * out_is;
                if (t < T) {
                    const float s_ = d1 + (r > 0 ? Of[192 + r - 1] : 0.f) + (r < NT - 1 ? Of[192 + NT + r + 1] : 0.f);
                    if (r == 0 || r == NT - 1) atomicAdd(lg + t, s_);
                    else lg[t] = s_;
                    if (r == NT - 1 && t + 1 < T) atomicAdd(lg + t + 1, d0);
                    if (r == 0 && t > 0) atomicAdd(lg + t - 1, d2);
                }
            } else {
#pragma unroll
            for (int k = 0; k < ROWS * VPR / 256; ++k) {
                const int idx = tid + k * 256, r = idx / VPR, v = idx % VPR;
                const int t = t0 + ps * ROWS + r;
                if (t < T)
                    *(nef_h8*)(yb + (size_t)t * COUT + v * 8) = *(const nef_h8*)(Ol + r * ORS + v * 16);
            }
            }
            __syncthreads();                 // staging read: the next pass / the next chunk's X may overwrite it
        }
    }
#undef PHW_FETCH
#undef PHW_STAGE
#undef PHW_A
}

// ------------------------------------------------------------------------------------------------------------
// hconv_pair_kernel: layers 1 and 2 in ONE pass for sequences of one tile (T <= 256, i.e. L <= 512 -- BASELINE config 4):
//   c1 = ReLU(conv(256 -> 128)(q * up2(x)) + b1)  stays in LDS as fp16 [258 rows][128 ch] (rows 0 / 257 = the conv's
//   zero padding), y = ReLU(conv(128 -> 128)(c1) + b2) reads its B operand straight from those rows: c1 is neither
//   written to nor read from memory (2 x 64 KB per pair), layer 2 needs no staging and no barrier inside its 24 k-steps.
//   block   = 512 threads = 8 waves (2 per SIMD), one pair at a time, persistent, 1 block per CU:
//             LDS = two X buffers (74 KB) + the c1 tile (70 KB)
//   wave    = 32 co x 128 t (four accumulator tiles) in both layers; 8 waves = 4 (co) x 2 (t)
//   A       = weight fragments from L2 in a ring of AD k-steps that runs on across chunks, layers and pairs
//   layer 1 = hconv_wide_kernel's scheme (double-buffered 64-channel chunks, staging pieces among the MFMAs); the
//             next pair's first chunk is fetched and staged during layer 2
//   y       = staged over the c1 rows after layer 2, whole 256-byte rows out; the stores drain under the next pair
// Same k order per output as the two-launch sequence and the same fp16 rounding of c1 -> bit-identical results.
// ------------------------------------------------------------------------------------------------------------
#define PHP_CRS 272   // bytes per c1 / y row in LDS: 128 halfs + 16 B pad (conflict-free ds_read_b128 across 16 rows)

template <bool TILED>
__global__ __launch_bounds__(512, 1) void hconv_pair_kernel(const _Float16* __restrict__ x, const nef_h8* __restrict__ wp1,
                                                         const float* __restrict__ bias1, const float* __restrict__ scale,
                                                         const nef_h8* __restrict__ wp2, const float* __restrict__ bias2,
                                                         _Float16* __restrict__ y, int T, int N, int x_div, int nq,
                                                         long sc_bs, long sc_is, int tiles_per_n) {
    // TILED (round 6; sequences longer than one tile -- configs[4]'s 2500 rows): a work item is (pair, tile); tile k produces the
    // NOUT = 252 output rows o = 252 k .. 252 k + 251 from 256-slot tiles of c1 and c2 whose slot j is time base + j, base = o - 2
    // (even: the x2 blend's row parity is that of the one-tile form); rows 0 / 257 of the X buffers hold real halo rows of the scaled,
    // upsampled input there and the conv's zero padding in the one-tile form (base = 0, slot = time).
    constexpr int NOUT = 252;
    constexpr int CIN = 256, COUT = 128, NT = 256, NI = 4;
    constexpr int XROWS = NT + 2;
    constexpr int XBYTES = XROWS * PH_XRS;
#ifndef NEF_PHP_AD
#define NEF_PHP_AD 6
#endif
    constexpr int AD = NEF_PHP_AD;           // A ring: fragments of AD - 1 k-steps in flight (a divisor of 12)
    static_assert(12 % AD == 0, "the ring position of a k-step must not depend on the chunk");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const C1 = smem + 2 * XBYTES;      // [258][PHP_CRS]: row r = time r - 1

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 3, wn = wave >> 2;  // wave = 32 co x 128 t (four accumulator tiles) in both layers: 8 waves = 4 (co) x 2 (t)
    const int lo = lane & 31, hi = lane >> 5;
    const int seg = tid & 7, rg = tid >> 3;  // staging: 8-channel segment, group of 4 output rows (0..63)
    const int Tin = T / 2;

    nef_f16acc acc[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
    nef_h8 hzero;
#pragma unroll
    for (int e = 0; e < 8; ++e) hzero[e] = (_Float16)0.f;

    // the padding rows of both X buffers and of the c1 tile are zero for the whole kernel (one tile per pair; TILED: the X buffers'
    // rows 0 / 257 are halo rows staged with every chunk -- by other threads than these, so zeroing them here would race)
    if (!TILED && tid < 32) {
        const int bsel = tid >> 4, r = ((tid >> 3) & 1) ? XROWS - 1 : 0;
        *(nef_h8*)(smem + bsel * XBYTES + r * PH_XRS + seg * 16) = hzero;
    }
    if (tid < 32) {
        const int r = (tid >> 4) ? XROWS - 1 : 0;
        *(nef_h8*)(C1 + r * PHP_CRS + (tid & 15) * 16) = hzero;
    }

    nef_h8 xr[4], qh;
    int fbase = 0;                            // `base` of the tile held in xr
    char* const st0 = smem + (4 * rg + 1) * PH_XRS + seg * 16;      // this thread's first staged row, buffer 0
    // source rows base / 2 + 2 rg - 1 .. + 2 (clamped to the sequence: the align_corners=False edge rule) of chunk cc_ of work item w_
#define PHP_FETCH(w_, cc_)                                                                                    \
    {                                                                                                         \
        const int n_ = TILED ? (w_) / tiles_per_n : (w_);                                                     \
        fbase = TILED ? ((w_) % tiles_per_n) * NOUT - 2 : 0;                                                  \
        const __amdgpu_buffer_rsrc_t xd_ = __builtin_amdgcn_make_buffer_rsrc(                                 \
            const_cast<_Float16*>(x + (size_t)((n_) / x_div) * Tin * CIN), 0, Tin * CIN * 2, 0x00020000);     \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                       \
            int r_ = fbase / 2 + 2 * rg - 1 + k;                                                              \
            r_ = r_ < 0 ? 0 : (r_ > Tin - 1 ? Tin - 1 : r_);                                                  \
            xr[k] = __builtin_bit_cast(nef_h8, __builtin_amdgcn_raw_buffer_load_b128(                         \
                xd_, r_ * (CIN * 2) + seg * 16, (cc_) * 128, 0));                                             \
        }                                                                                                     \
        const float* sc_ = scale + (size_t)((n_) / nq) * sc_bs + (size_t)((n_) % nq) * sc_is + (cc_) * 64;    \
        const __amdgpu_buffer_rsrc_t sd_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sc_), 0, 256, 0x00020000); \
        const nef_f32x4 q0_ = nef_buf_f32x4(sd_, seg * 32, 0), q1_ = nef_buf_f32x4(sd_, seg * 32 + 16, 0);    \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) { qh[e] = (_Float16)q0_[e]; qh[4 + e] = (_Float16)q1_[e]; } \
    }
    // blended, scaled rows j (of this thread's 4: time fbase + 4 rg + j) -> X buffer BUF_; same arithmetic as hconv_wide_kernel.  Rows
    // outside the sequence are not masked here (four selects per row, on every row of every pair, for rows that only a ragged last
    // tile has): PHP_PAD_ROWS zeroes the two that matter -- times -1 and T, the conv's padding -- afterwards.  TILED, j = 4
    // (J1_ = 5): the two halo rows 0 / 257 (times fbase - 1, fbase + 256) by the first / last row group
#define PHP_STAGE(BUF_, J0_, J1_)                                                                             \
    {                                                                                                         \
        nef_h8 c75;                                                                                           \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) c75[e] = (_Float16)0.75f;                               \
        _Pragma("unroll") for (int j = (J0_); j < (J1_) && j < 4; ++j) {                                      \
            const nef_h8 a_ = xr[(j >> 1) + 1], b_ = (j & 1) ? xr[(j >> 1) + 2] : xr[j >> 1];                 \
            const nef_h8 v_ = __builtin_elementwise_fma(a_, c75, b_ * (_Float16)0.25f) * qh;                  \
            *(nef_h8*)(st0 + (BUF_) * XBYTES + j * PH_XRS) = v_;                                              \
        }                                                                                                     \
        if (TILED && (J1_) > 4 && (rg == 0 || rg == 63)) {                                                    \
            const nef_h8 a_ = rg == 0 ? xr[0] : xr[3], b_ = rg == 0 ? xr[1] : xr[2];                          \
            nef_h8 v_ = __builtin_elementwise_fma(a_, c75, b_ * (_Float16)0.25f) * qh;                        \
            const int tau_ = rg == 0 ? fbase - 1 : fbase + NT;                                                \
            if (tau_ >= T || tau_ < 0) v_ = hzero;                                                            \
            *(nef_h8*)(smem + (BUF_) * XBYTES + (rg == 0 ? 0 : XROWS - 1) * PH_XRS + seg * 16) = v_;          \
        }                                                                                                     \
    }
    // rows of the staged tile (base fb_) that hold times -1 and T.  Block-uniform and rare (ragged or first tiles), so the extra
    // barrier -- the rows were stored by other threads -- costs nothing in the common case.
#define PHP_PAD_ROWS(BUF_, fb_)                                                                               \
    if ((fb_) + NT > T || (TILED && (fb_) < 0)) {                                                             \
        __syncthreads();                                                                                      \
        if (tid < 16) {                                                                                       \
            const int r_ = (tid >> 3) ? T - (fb_) + 1 : -(fb_);      /* row r holds time fb_ + r - 1 */       \
            if (r_ >= (TILED ? 0 : 1) && r_ <= NT + 1) *(nef_h8*)(smem + (BUF_) * XBYTES + r_ * PH_XRS + seg * 16) = hzero; \
        }                                                                                                     \
    }
    const __amdgpu_buffer_rsrc_t wd1 = nef_rsrc(wp1), wd2 = nef_rsrc(wp2);
    const int avoff = lane * 16 + wm * 1024;
    nef_h8 a[AD];
    // A fragment of k-step `stage_` (four 1 KB fragments per k-step = 128 output channels; this wave's 32 are fragment wm)
#define PHP_A(wd_, slot_, stage_)                                                                             \
    a[slot_] = __builtin_bit_cast(nef_h8, __builtin_amdgcn_raw_buffer_load_b128(wd_, avoff, (stage_) * 4096, 0));
    // 12 k-steps (3 taps x 4 x 16 channels) of one 64-channel chunk: B fragments at Bp_ + (ni*32 + tap) * PITCH_ + kq*32, read a whole
    // k-step ahead; A of k-step s + AD - 1 comes from (WDC_, stage SC_ + .) or, past the chunk, from (WDN_, stage SN_ + .);
    // FE_: fetch (pair, chunk) for the staging at k-step 1; ST_: stage into buffer BUF_ during k-steps 6..9
#define PHP_STEPS(Bp_, PITCH_, WDC_, SC_, WDN_, SN_, FE_, FN_, FC_, ST_, BUF_)                                \
    {                                                                                                         \
        nef_h8 b[2][NI];                                                                                      \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) b[0][ni] = *(const nef_h8*)((Bp_) + ni * 32 * (PITCH_)); \
        _Pragma("unroll") for (int s = 0; s < 12; ++s) {                                                      \
            if (s + AD - 1 < 12) {                                                                            \
                PHP_A(WDC_, (s + AD - 1) % AD, (SC_) + s + AD - 1)                                            \
            } else {                                                                                          \
                PHP_A(WDN_, (s + AD - 1) % AD, (SN_) + s + AD - 1 - 12)                                       \
            }                                                                                                 \
            if (s + 1 < 12)                                                                                   \
                _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                             \
                    b[(s + 1) & 1][ni] = *(const nef_h8*)((Bp_) + (ni * 32 + (s + 1) / 4) * (PITCH_) + ((s + 1) % 4) * 32); \
            if ((FE_) && s == 1) PHP_FETCH(FN_, FC_)                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                                \
            if ((ST_) && s >= 6 && s <= 9) PHP_STAGE(BUF_, s - 6, s - 5)                                      \
            if ((ST_) && TILED && s == 10) PHP_STAGE(BUF_, 4, 5)                                              \
            _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                                 \
                acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s % AD], b[s & 1][ni], acc[ni], 0, 0, 0);  \
            _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) {                                               \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                            \
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                                            \
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                            \
            }                                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                \
        }                                                                                                     \
    }
    // accumulators -> bias + ReLU -> fp16 rows of the c1 region (row 1 + slot): two packed adds, two packed converts (round to nearest
    // even, as v_cvt_f16_f32) and two packed max per four values (rounds 2-5: 19 vector instructions per four)
#define PHP_TO_LDS(bv_)                                                                                       \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                           \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) {                                                   \
            nef_f32x4 v_;                                                                                     \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                   \
                v_[e] = acc[ni][g * 4 + e];                                                                   \
                acc[ni][g * 4 + e] = 0.f;                                                                     \
            }                                                                                                 \
            const nef_h4 z_ = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};                   \
            *(nef_h4*)(eb + ni * 32 * PHP_CRS + g * 16) =                                                     \
                __builtin_elementwise_max(__builtin_convertvector(v_ + (bv_)[g], nef_h4), z_);                \
        }                                                                                                     \
    }

    // the biases of this wave's 32 output channels, once (a load inside the epilogue would also wait -- memory returns in order -- for
    // whatever was issued ahead of it)
    nef_f32x4 bv1[4], bv2[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        bv1[g] = *(const nef_f32x4*)(bias1 + wm * 32 + 8 * g + 4 * hi);
        bv2[g] = *(const nef_f32x4*)(bias2 + wm * 32 + 8 * g + 4 * hi);
    }
    char* const eb = C1 + (1 + wn * 128 + lo) * PHP_CRS + (wm * 32 + 4 * hi) * 2;      // epilogue rows: + ni 32 rows + g 16 bytes
    const int total = TILED ? N * tiles_per_n : N;      // work items: (pair, tile)
    int w = blockIdx.x;
#pragma unroll
    for (int j = 0; j < AD - 1; ++j) PHP_A(wd1, j, j)
    if (w < total) {
        PHP_FETCH(w, 0)
        PHP_STAGE(0, 0, 5)
        PHP_PAD_ROWS(0, fbase)
    }
    __syncthreads();

#pragma unroll 1
    for (; w < total; w += gridDim.x) {
        const int n = TILED ? w / tiles_per_n : w;
        const int base = TILED ? (w % tiles_per_n) * NOUT - 2 : 0;      // time of slot 0
        const int n_next = w + (int)gridDim.x < total ? w + (int)gridDim.x : w;   // last item: re-stage itself (unread)
        // ---- layer 1: four 64-channel chunks, buffers 0 1 0 1; chunk cc + 1 staged during chunk cc
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const char* Bp = smem + (cc & 1) * XBYTES + (wn * 128 + lo) * PH_XRS + 16 * hi;
            if (cc < 3) {
                PHP_STEPS(Bp, PH_XRS, wd1, cc * 12, wd1, (cc + 1) * 12, true, w, cc + 1, true, (cc + 1) & 1)
                PHP_PAD_ROWS((cc + 1) & 1, base)
            } else {
                PHP_STEPS(Bp, PH_XRS, wd1, cc * 12, wd2, 0, false, w, 0, false, 0)
            }
            __syncthreads();
        }
        PHP_TO_LDS(bv1)
        // slots of c1 outside the sequence are the next conv's zero padding (block-uniform: ragged / first tiles only)
        if (base + NT > T || (TILED && base < 0)) {
            __syncthreads();
#pragma unroll 1
            for (int r_ = tid >> 4; r_ < NT; r_ += 32)
                if (base + r_ >= T || base + r_ < 0) *(nef_h8*)(C1 + (1 + r_) * PHP_CRS + (tid & 15) * 16) = hzero;
        }
        __syncthreads();                     // c1 complete
        // ---- layer 2: two 64-channel chunks straight from the c1 rows; the next pair's first X chunk rides along
        {
            const char* Bp = C1 + (wn * 128 + lo) * PHP_CRS + 16 * hi;
            PHP_STEPS(Bp, PHP_CRS, wd2, 0, wd2, 12, false, w, 0, false, 0)
            PHP_STEPS(Bp + 128, PHP_CRS, wd2, 12, wd1, 0, true, n_next, 0, true, 0)
            PHP_PAD_ROWS(0, fbase)           // (fbase: the next item's tile)
        }
        __syncthreads();                     // every wave is done reading c1; X chunk 0 of the next pair is staged
        PHP_TO_LDS(bv2)
        __syncthreads();
        _Float16* yb = y + (size_t)n * T * COUT;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int idx = tid + k * 512, r = idx >> 4, v = idx & 15;
            // slot r is time base + r; a tile of the TILED form owns slots 2 .. 253 (the others lack context)
            if (base + r < T && (!TILED || (r >= 2 && r < 2 + NOUT)))
                *(nef_h8*)(yb + (size_t)(base + r) * COUT + v * 8) = *(const nef_h8*)(C1 + (1 + r) * PHP_CRS + v * 16);
        }
        // no barrier: the c1 rows are next written four chunk barriers from here
    }
#undef PHP_FETCH
#undef PHP_STAGE
#undef PHP_PAD_ROWS
#undef PHP_A
#undef PHP_STEPS
#undef PHP_TO_LDS
}

// ------------------------------------------------------------------------------------------------------------
// hconv_tail_kernel (round 6): layers 3 and 4, the last conv and sigmoid(x/3) in ONE pass for sequences of one tile
// (T <= 512 output rows, i.e. L <= 512 -- BASELINE config 4):
//   c3 = ReLU(conv(128 -> 64)(up2(c2)) + b3)  and  c4 = ReLU(conv(64 -> 64)(c3) + b4)  never leave the chip; the 64 -> 1 conv is
//   evaluated on the staged c4 rows and the finished fp32 view is the only thing written.  Per (sample, angle) pair the two-launch
//   sequence wrote and read 64 KB of c3 and took two more small launches (edge zeroing, sigmoid); this reads c2 once (64 KB) and
//   writes 2 KB.
//   block   = 512 threads = 8 waves along time (64 co x 64 t each: 2 x 2 accumulator tiles), one pair at a time, persistent,
//             1 block per CU.  LDS = two row buffers of 514 rows x 144 B (64 channels + pad): buffer A holds layer 3's first input
//             chunk, buffer B its second chunk, THEN c3 (written by layer 3's epilogue over the consumed chunk), THEN c4.
//   layer 3 = hconv_pair_kernel's scheme with 8-row staging groups (hconv_wide_kernel's x2 blend: 6 source rows -> 8 rows): chunk 1
//             is fetched and staged during chunk 0's 12 k-steps
//   layer 4 = 12 k-steps straight from the c3 rows, no staging and no barrier inside; the NEXT pair's first chunk is fetched and
//             staged into buffer A meanwhile
//   last conv = on the matrix cores too (a 3-row A tile of the fp32 weights as two fp16 terms against the staged c4 rows), the three
//             taps' partial sums meet through LDS; the whole sequence sits in the block, so there are no tile edges, no atomics and
//             no separate sigmoid pass
// Same k order per output and the same fp16 roundings of c3 / c4 as the launches it replaces; the last conv's 64-term sums run in the
// matrix cores' order instead of one fused-multiply-add chain (views agree to fp32 round-off: tests/test_pano_gpu.py).
// ------------------------------------------------------------------------------------------------------------
template <bool TILED>
__global__ __launch_bounds__(512, 1) void hconv_tail_kernel(const _Float16* __restrict__ x, const nef_h8* __restrict__ wp3,
                                                         const float* __restrict__ bias3, const nef_h8* __restrict__ wp4,
                                                         const float* __restrict__ bias4, const float* __restrict__ wout,
                                                         const float* __restrict__ bout, float* __restrict__ out, int T, int N,
                                                         int nq, long out_bs, long out_is, int tiles_per_n) {
    // TILED (sequences longer than one tile -- configs[4]'s 5000 rows): a work item is (pair, tile); tile k produces the NOUT = 508
    // output rows o = 508 k .. 508 k + 507 from 512-slot tiles of c3 and c4 whose slot j is time base + j, base = o - 2: two fused
    // K = 3 layers + the last conv eat three rows of context per side, of which the row buffers' rows 0 / 513 supply one -- as real
    // halo rows of the upsampled input here, as the convs' zero padding in the one-tile form (base = 0, slot = time).
    constexpr int NOUT = 508;
    constexpr int CIN = 128, NT = 512, NI = 4;
    constexpr int XROWS = NT + 2;
    constexpr int XBYTES = XROWS * PH_XRS;
#ifndef NEF_PHT_AD
#define NEF_PHT_AD 6
#endif
#ifndef NEF_PHT_EARLY
#define NEF_PHT_EARLY 1  // the input rows of a staging pass are fetched one phase ahead, just before the preceding epilogue (see the main loop)
#endif
    constexpr int AD = NEF_PHT_AD;          // A ring: fragments of AD - 1 k-steps in flight (a divisor of 12; one 1 KB fragment per k-step and wave: 6 = 24 registers)
    static_assert(12 % AD == 0, "the ring position of a k-step must not depend on the chunk");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const XA = smem;
    char* const XB = smem + XBYTES;           // layer 3's second chunk, then c3, then c4: row r = time r - 1
    float* const Of = (float*)(smem + 2 * XBYTES);      // d0[NT], d2[NT], d1[NT]
    nef_h8* const Af = (nef_h8*)(smem + 2 * XBYTES + 3 * NT * 4);      // the last conv as matrix A fragments: [kq 0..3][hi | lo plane][lane]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;  // wave = 32 co x 128 t: 8 waves = 2 (co) x 4 (t)
    const int lo = lane & 31, hi = lane >> 5;
    const int seg = tid & 7, rg = tid >> 3;  // staging: 8-channel segment, group of 8 output rows (0..63)
    const int Tin = T / 2;
    // The 64 -> 1 conv on the matrix cores: d_k[t] = sum_c wout[c][k] c4[t][c] is a [3 x 64] x [64 x 512] product -- rows 0..2 of a 32-row
    // A tile (the rest zero), the c4 rows as B exactly as layer 4 reads c3.  wout stays fp32-exact: it enters as two fp16 terms
    // (hi + lo = 22 bits), two matrix instructions per 16 channels.  16 matrix instructions per wave replace ~400 vector
    // instructions per thread (one thread per row, 192 fused multiply-adds) during which the matrix pipes of this one-block-per-CU
    // kernel stood still.
    if (tid < 64) {
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            nef_h8 fh, fl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = kq * 16 + 8 * hi + e;
                const float w_ = lo < 3 ? wout[c * 3 + lo] : 0.f;
                fh[e] = (_Float16)w_;
                fl[e] = (_Float16)(w_ - (float)fh[e]);
            }
            Af[(kq * 2 + 0) * 64 + lane] = fh;
            Af[(kq * 2 + 1) * 64 + lane] = fl;
        }
    }

    nef_f16acc acc[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
    nef_h8 hzero;
#pragma unroll
    for (int e = 0; e < 8; ++e) hzero[e] = (_Float16)0.f;
    // one-tile form: rows 0 and NT + 1 of both buffers are the convs' zero padding for the whole kernel (TILED: they are halo rows,
    // staged with every chunk -- by OTHER threads than these, so zeroing them here would race with the first item's staging)
    if (!TILED && tid < 32) {
        const int bsel = tid >> 4, r = ((tid >> 3) & 1) ? XROWS - 1 : 0;
        *(nef_h8*)(smem + bsel * XBYTES + r * PH_XRS + seg * 16) = hzero;
    }

    nef_h8 xr[6];
    int fbase = 0;                            // `base` of the tile held in xr
    char* const stA = XA + (8 * rg + 1) * PH_XRS + seg * 16;      // this thread's first staged row
    char* const stB = stA + XBYTES;
    // rows of the staged tile (base fb_) that hold times -1 and T: the convs' zero padding.  Block-uniform and rare (ragged or first
    // tiles), so the extra barrier -- the stores above come from other threads -- is free in the common case.
#define PHT_PAD_ROWS(Xn_, fb_)                                                                                \
    if ((fb_) + NT > T || (TILED && (fb_) < 0)) {                                                            \
        __syncthreads();                                                                                      \
        if (tid < 16) {                                                                                       \
            const int r_ = (tid >> 3) ? T - (fb_) + 1 : -(fb_);      /* row r holds time fb_ + r - 1 */       \
            if (r_ >= (TILED ? 0 : 1) && r_ <= NT + 1) *(nef_h8*)((Xn_) + r_ * PH_XRS + seg * 16) = hzero;    \
        }                                                                                                     \
    }
    // source rows base / 2 + 4 rg - 1 .. + 4 (clamped: the align_corners=False edge rule) of the 64-channel chunk cc_ of work item w_
#define PHT_FETCH(w_, cc_)                                                                                    \
    {                                                                                                         \
        const int n_ = TILED ? (w_) / tiles_per_n : (w_);                                                     \
        fbase = TILED ? ((w_) % tiles_per_n) * NOUT - 2 : 0;                                                  \
        const __amdgpu_buffer_rsrc_t xd_ = __builtin_amdgcn_make_buffer_rsrc(                                 \
            const_cast<_Float16*>(x + (size_t)(n_) * Tin * CIN), 0, Tin * CIN * 2, 0x00020000);               \
        _Pragma("unroll") for (int k = 0; k < 6; ++k) {                                                       \
            int r_ = fbase / 2 + 4 * rg - 1 + k;                                                              \
            r_ = r_ < 0 ? 0 : (r_ > Tin - 1 ? Tin - 1 : r_);                                                  \
            xr[k] = __builtin_bit_cast(nef_h8, __builtin_amdgcn_raw_buffer_load_b128(                         \
                xd_, r_ * (CIN * 2) + seg * 16, (cc_) * 128, 0));                                             \
        }                                                                                                     \
    }
    // blended rows j (of this thread's 8: time fbase + 8 rg + j) -> row buffer; hconv_wide_kernel's arithmetic (0.25 b exact, one fma
    // rounding).  Rows outside the sequence are NOT masked here (four selects per row on every row of every pair, for rows that exist
    // only in a ragged last tile): PHT_PAD_ROWS zeroes the two rows that matter -- times -1 and T, the conv's padding -- afterwards.
    // TILED, j = 8 (J1_ = 9): the two halo rows 0 / 513 (times fbase - 1, fbase + 512) by the first / last row group
#define PHT_STAGE(Xn_, J0_, J1_)                                                                              \
    {                                                                                                         \
        nef_h8 c75;                                                                                           \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) c75[e] = (_Float16)0.75f;                               \
        _Pragma("unroll") for (int j = (J0_); j < (J1_) && j < 8; ++j) {                                      \
            const nef_h8 a_ = xr[(j >> 1) + 1], b_ = (j & 1) ? xr[(j >> 1) + 2] : xr[j >> 1];                 \
            const nef_h8 v_ = __builtin_elementwise_fma(a_, c75, b_ * (_Float16)0.25f);                       \
            *(nef_h8*)(((Xn_) == XA ? stA : stB) + j * PH_XRS) = v_;                                          \
        }                                                                                                     \
        if (TILED && (J1_) > 8 && (rg == 0 || rg == 63)) {                                                    \
            const nef_h8 a_ = rg == 0 ? xr[0] : xr[5], b_ = rg == 0 ? xr[1] : xr[4];                          \
            nef_h8 v_ = __builtin_elementwise_fma(a_, c75, b_ * (_Float16)0.25f);                             \
            const int tau_ = rg == 0 ? fbase - 1 : fbase + NT;                                                \
            if (tau_ >= T || tau_ < 0) v_ = hzero;                                                            \
            *(nef_h8*)((Xn_) + (rg == 0 ? 0 : XROWS - 1) * PH_XRS + seg * 16) = v_;                           \
        }                                                                                                     \
    }
    const __amdgpu_buffer_rsrc_t wd3 = nef_rsrc(wp3), wd4 = nef_rsrc(wp4);
    const int avoff = lane * 16 + wm * 1024;
    nef_h8 a[AD];
    // A fragment of k-step `stage_` (two 1 KB fragments per k-step = 64 output channels; this wave's 32 are fragment wm)
#define PHT_A(wd_, slot_, stage_)                                                                             \
    a[slot_] = __builtin_bit_cast(nef_h8, __builtin_amdgcn_raw_buffer_load_b128(wd_, avoff, (stage_) * 2048, 0));
    // 12 k-steps (3 taps x 4 x 16 channels) of one 64-channel chunk, B fragments at Bp_ + (ni * 32 + tap) * PH_XRS + kq * 32;
    // A of k-step s + AD - 1 from (WDC_, SC_ + .) or, past the chunk, from (WDN_, SN_ + .); FE_: fetch (pair FN_, chunk FC_) at
    // k-step 1; ST_: stage the fetched rows into XN_, one row per k-step, during k-steps 4 .. 11
#define PHT_STEPS(Bp_, WDC_, SC_, WDN_, SN_, FE_, FN_, FC_, ST_, XN_)                                         \
    {                                                                                                         \
        nef_h8 b[2][NI];                                                                                      \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) b[0][ni] = *(const nef_h8*)((Bp_) + ni * 32 * PH_XRS); \
        _Pragma("unroll") for (int s = 0; s < 12; ++s) {                                                      \
            if (s + AD - 1 < 12) {                                                                            \
                PHT_A(WDC_, (s + AD - 1) % AD, (SC_) + s + AD - 1)                                            \
            } else {                                                                                          \
                PHT_A(WDN_, (s + AD - 1) % AD, (SN_) + s + AD - 1 - 12)                                       \
            }                                                                                                 \
            /* the B fragments of k-step s + 1 go out a whole k-step ahead (behind the staging store they were read one matrix */ \
            /* instruction ahead and every k-step opened on a full LDS round trip) */                        \
            if (s + 1 < 12)                                                                                   \
                _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                             \
                    b[(s + 1) & 1][ni] = *(const nef_h8*)((Bp_) + (ni * 32 + (s + 1) / 4) * PH_XRS + ((s + 1) % 4) * 32); \
            if ((FE_) && !NEF_PHT_EARLY && s == 1) PHT_FETCH(FN_, FC_)                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                \
            if ((ST_) && s >= 4) PHT_STAGE(XN_, s - 4, s - 3)                                                 \
            if ((ST_) && TILED && s == 11) PHT_STAGE(XN_, 8, 9)                                               \
            _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) {                                               \
                acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s % AD], b[s & 1][ni], acc[ni], 0, 0, 0);  \
            }                                                                                                 \
            _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) {                                               \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                            \
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);                                            \
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                            \
            }                                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                \
        }                                                                                                     \
    }
    // accumulators -> bias + ReLU -> fp16 rows of buffer B (row 1 + slot): two packed adds, two packed converts (round to nearest even,
    // as v_cvt_f16_f32) and two packed max per four values -- round 6 first half: 19 vector instructions per four (per-value add, max,
    // convert, select, pack).  Slots outside the sequence (the next conv's zero padding, zero weight in the last conv) are zeroed
    // afterwards by PHT_ZERO_SLOTS instead of selected per value.
#define PHT_TO_LDS(bv_)                                                                                       \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                           \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) {                                                   \
            nef_f32x4 v_;                                                                                     \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                   \
                v_[e] = acc[ni][g * 4 + e];                                                                   \
                acc[ni][g * 4 + e] = 0.f;                                                                     \
            }                                                                                                 \
            const nef_h4 z_ = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};                   \
            *(nef_h4*)(eb + ni * 32 * PH_XRS + g * 16) =                                                      \
                __builtin_elementwise_max(__builtin_convertvector(v_ + (bv_)[g], nef_h4), z_);                \
        }                                                                                                     \
    }
    // slots t of buffer B with base + t outside [0, T): rows 1 + t -> 0 (block-uniform, ragged / first tiles only)
#define PHT_ZERO_SLOTS()                                                                                      \
    if (base + NT > T || (TILED && base < 0)) {                                                               \
        __syncthreads();                                                                                      \
        _Pragma("unroll 1") for (int r_ = rg; r_ < NT; r_ += 64)                                              \
            if (base + r_ >= T || base + r_ < 0) *(nef_h8*)(XB + (1 + r_) * PH_XRS + seg * 16) = hzero;       \
    }

    // the biases of this wave's 32 output channels, once (a load inside the epilogue would also wait -- memory returns in order -- for
    // the input rows fetched just ahead of it)
    nef_f32x4 bv3[4], bv4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        bv3[g] = *(const nef_f32x4*)(bias3 + wm * 32 + 8 * g + 4 * hi);
        bv4[g] = *(const nef_f32x4*)(bias4 + wm * 32 + 8 * g + 4 * hi);
    }
    char* const eb = XB + (1 + wn * 128 + lo) * PH_XRS + (wm * 32 + 4 * hi) * 2;      // epilogue rows: + ni 32 rows + g 16 bytes
    const int total = TILED ? N * tiles_per_n : N;      // work items: (pair, tile)
    int w = blockIdx.x;
#pragma unroll
    for (int j = 0; j < AD - 1; ++j) PHT_A(wd3, j, j)
    if (w < total) {
        PHT_FETCH(w, 0)
        PHT_STAGE(XA, 0, 9)
        PHT_PAD_ROWS(XA, fbase)
        if (NEF_PHT_EARLY) PHT_FETCH(w, 1)
    }
    __syncthreads();
    const float b0 = bout[0];

#pragma unroll 1
    for (; w < total; w += gridDim.x) {
        const int n = TILED ? w / tiles_per_n : w;
        const int base = TILED ? (w % tiles_per_n) * NOUT - 2 : 0;      // time of slot 0
        const int n_next = w + (int)gridDim.x < total ? w + (int)gridDim.x : w;   // last item: re-stage itself (unread)
        const char* const BA = XA + (wn * 128 + lo) * PH_XRS + 16 * hi;
        const char* const BB = XB + (wn * 128 + lo) * PH_XRS + 16 * hi;
        // ---- layer 3: chunk 0 from buffer A while chunk 1 is fetched and staged into buffer B, then chunk 1
        PHT_STEPS(BA, wd3, 0, wd3, 12, true, w, 1, true, XB)
        PHT_PAD_ROWS(XB, base)
        __syncthreads();
        PHT_STEPS(BB, wd3, 12, wd4, 0, false, w, 0, false, XB)
        // Memory returns in issue order: a weight fragment issued behind these rows cannot be consumed before they have arrived.  Issued
        // here the rows have the epilogue and AD - 1 k-steps to come from HBM; issued at k-step 1 of the staging phase (rounds 2-5)
        // they had AD k-steps, and the matrix pipes waited for them.
        if (NEF_PHT_EARLY) PHT_FETCH(n_next, 0)
        __syncthreads();                     // every wave is done reading buffer B as an input chunk
        PHT_TO_LDS(bv3)
        PHT_ZERO_SLOTS()
        __syncthreads();                     // c3 complete
        // ---- layer 4: straight from the c3 rows; the next pair's first chunk rides along into buffer A
        PHT_STEPS(BB, wd4, 0, wd3, 0, true, n_next, 0, true, XA)
        PHT_PAD_ROWS(XA, fbase)              // (fbase: the next item's tile)
        if (NEF_PHT_EARLY) PHT_FETCH(n_next, 1)      // for the next item's first phase
        __syncthreads();                     // every wave is done reading c3; buffer A holds the next pair's chunk 0
        PHT_TO_LDS(bv4)
        PHT_ZERO_SLOTS()                     // (rows outside the sequence zero: their share of the last conv is zero)
        __syncthreads();                     // c4 staged
        // ---- last conv on the matrix cores: rows 0..2 of the product are d0, d1, d2 of column t (lanes hi = 0); the two waves of
        // a 128-column stripe take 64 columns each: t = wn 128 + wm 64 + ni 32 + lo.  The hi and lo weight terms go to separate
        // accumulators (no matrix instruction waits for the one before it) and meet in the d arrays.
        {
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                const nef_h8 ah = Af[(kq * 2 + 0) * 64 + lane], al = Af[(kq * 2 + 1) * 64 + lane];
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const nef_h8 bq = *(const nef_h8*)(XB + (1 + wn * 128 + wm * 64 + ni * 32 + lo) * PH_XRS + kq * 32 + 16 * hi);
                    acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bq, acc[ni], 0, 0, 0);
                    acc[2 + ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bq, acc[2 + ni], 0, 0, 0);
                }
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int t = wn * 128 + wm * 64 + ni * 32 + lo;
                if (hi == 0) {
                    Of[t] = acc[ni][0] + acc[2 + ni][0];                  // tap 0 weights this row into column t + 1
                    Of[NT + t] = acc[ni][2] + acc[2 + ni][2];             // tap 2 into column t - 1
                    Of[2 * NT + t] = acc[ni][1] + acc[2 + ni][1];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ni][r] = acc[2 + ni][r] = 0.f;
            }
            __syncthreads();           // (also: every wave is done reading c4 -- buffer B is free for the next pair's chunk 1)
            {
                const int t = wn * 128 + wm * 64 + lane;      // one column per lane, whole 256-byte rows out
                // slot t is time base + t; a tile of the TILED form owns slots 2 .. 509 (the others lack context)
                if (base + t < T && (!TILED || (t >= 2 && t < 2 + NOUT))) {
                    const float s_ = Of[2 * NT + t] + (t > 0 ? Of[t - 1] : 0.f) + (t < NT - 1 ? Of[NT + t + 1] : 0.f);
                    out[(size_t)(n / nq) * out_bs + (size_t)(n % nq) * out_is + base + t] = 1.0f / (1.0f + expf(-(s_ + b0) / 3.0f));
                }
            }
            // the d arrays are rewritten five barriers from here
        }
    }
#undef PHT_FETCH
#undef PHT_STAGE
#undef PHT_A
#undef PHT_STEPS
#undef PHT_TO_LDS
#undef PHT_ZERO_SLOTS
#undef PHT_PAD_ROWS
}

// ------------------------------------------------------------------------------------------------------------
// Last conv 64 -> 1 (k3, bias) + sigmoid(x/3) (model_nefnet.py:106,:168/:186): HBM-bound.  8 lanes share one time
// step (8 channels x 3 taps each), a block covers 256 consecutive time steps of one pair.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ph_outconv_kernel(const _Float16* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ out, int T,
                                                         int tiles_per_n, int nq, long out_bs, long out_is) {
    __shared__ float res[256];
    const int tid = threadIdx.x, seg = tid & 7, row = tid >> 3;
    const int n = blockIdx.x / tiles_per_n, t0 = (blockIdx.x % tiles_per_n) * 256;
    float wr[3][8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int k = 0; k < 3; ++k) wr[k][e] = w[(seg * 8 + e) * 3 + k];
    const float b0 = bias[0];
    const _Float16* xb = x + (size_t)n * T * 64;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int t = t0 + p * 32 + row;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int tt = t + k - 1;
            if (tt >= 0 && tt < T && t < T) {
                const nef_h8 v = *(const nef_h8*)(xb + (size_t)tt * 64 + seg * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += wr[k][e] * (float)v[e];
            }
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        if (seg == 0) res[p * 32 + row] = s;
    }
    __syncthreads();
    const int t = t0 + tid;
    if (t < T) {
        const float v = (res[tid] + b0) / 3.0f;
        out[(size_t)(n / nq) * out_bs + (size_t)(n % nq) * out_is + t] = 1.0f / (1.0f + expf(-v));
    }
}

// ------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------
// zero the tile-edge columns the fused kernel accumulates into (t = k*NT and k*NT + NT-1)
__global__ void ph_zero_edges_kernel(float* __restrict__ logit, int N, int T, int NT, int tiles_per_n, int nq,
                                     long out_bs, long out_is) {
    const int64_t total = (int64_t)N * tiles_per_n * 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int side = (int)(i & 1);
        const int tile = (int)((i >> 1) % tiles_per_n);
        const int n = (int)((i >> 1) / tiles_per_n);
        const int t = tile * NT + (side ? NT - 1 : 0);
        if (t < T) logit[(size_t)(n / nq) * out_bs + (size_t)(n % nq) * out_is + t] = 0.f;
    }
}

// out = sigmoid((logit + bias) / 3) in place (model_nefnet.py:168/:186)
__global__ void ph_sigmoid3_kernel(float* __restrict__ io, const float* __restrict__ bias, int N, int T, int nq,
                                   long out_bs, long out_is) {
    const float b0 = bias[0];
    const int64_t total = (int64_t)N * T;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / T), t = (int)(i % T);
        float* p = io + (size_t)(n / nq) * out_bs + (size_t)(n % nq) * out_is + t;
        *p = 1.0f / (1.0f + expf(-(*p + b0) / 3.0f));
    }
}

template <int CIN, int COUT, int PRO, int OUT = 0, int NI = 2, int MINB = 2>
static int launch_hconv(const void* x, const void* wp, const float* bias, const float* scale, void* y, int N, int T,
                        int x_div, int nq, long sc_bs, long sc_is, hipStream_t st, const float* wout = nullptr,
                        float* logit = nullptr, long out_bs = 0, long out_is = 0) {
    constexpr int NT = (4 / (COUT / 64)) * NI * 32;
    constexpr int XB = (NT + 2) * PH_XRS;
    constexpr int SB = (PRO & 2) ? (NT / 2 + 4) * PH_XRS : 0;
    constexpr int LDS = XB + SB + 3 * COUT * 64 * 2 + (OUT ? (192 + 2 * NT) * 4 : 0);
    const int tiles = (T + NT - 1) / NT;
    const int64_t total = (int64_t)N * tiles;
    if (total > 0x7FFFFFFF) return NEF_E_SHAPE;
    auto k = hconv_kernel<CIN, COUT, PRO, OUT, NI, MINB>;
    // blocks that fit the device at once (persistent grid): per device, idempotent -> thread-safe without a lock
    static int resident_dev[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    int resident = __atomic_load_n(&resident_dev[dev & 63], __ATOMIC_ACQUIRE);
    if (resident == 0) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        int cus = 0, per_cu = 0;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return (int)e;
        if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k, 256, LDS)) != hipSuccess)
            return (int)e;
        resident = cus * (per_cu > 0 ? per_cu : 1);
        __atomic_store_n(&resident_dev[dev & 63], resident, __ATOMIC_RELEASE);
    }
    const int grid = (int)(total < resident ? total : resident);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), LDS, st, (const _Float16*)x, (const nef_h8*)wp, bias, scale,
                       (_Float16*)y, T, tiles, (int)total, x_div, nq, sc_bs, sc_is, wout, logit, out_bs, out_is);
    return nef_launch_status();
}

template <int CIN, int COUT, int PRO, int OUT = 0>
static int launch_hconv_wide(const void* x, const void* wp, const float* bias, const float* scale, void* y, int N, int T,
                             int x_div, int nq, long sc_bs, long sc_is, hipStream_t st, const float* wout = nullptr,
                             float* logit = nullptr, long out_bs = 0, long out_is = 0) {
    constexpr int NT = 256;
    constexpr int LDS = 2 * (NT + 2) * PH_XRS + 256 * 16 + (OUT ? (192 + 2 * NT) * 4 : 0);
    const int tiles = (T + NT - 1) / NT;
    const int64_t total = (int64_t)N * tiles;
    if (total > 0x7FFFFFFF) return NEF_E_SHAPE;
    auto k = hconv_wide_kernel<CIN, COUT, PRO, OUT>;
    static int resident_dev[64] = {0};       // per device, idempotent -> thread-safe without a lock
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    int resident = __atomic_load_n(&resident_dev[dev & 63], __ATOMIC_ACQUIRE);
    if (resident == 0) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        int cus = 0, per_cu = 0;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return (int)e;
        if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k, 256, LDS)) != hipSuccess)
            return (int)e;
        resident = cus * (per_cu > 0 ? per_cu : 1);
        __atomic_store_n(&resident_dev[dev & 63], resident, __ATOMIC_RELEASE);
    }
    const int grid = (int)(total < resident ? total : resident);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), LDS, st, (const _Float16*)x, (const nef_h8*)wp, bias, scale,
                       (_Float16*)y, T, tiles, (int)total, x_div, nq, sc_bs, sc_is, wout, logit, out_bs, out_is);
    return nef_launch_status();
}

static int launch_hconv_pair(const void* x, const void* wp1, const float* b1, const float* scale, const void* wp2,
                             const float* b2, void* y, int N, int T, int x_div, int nq, long sc_bs, long sc_is,
                             hipStream_t st) {
    constexpr int LDS = 2 * 258 * PH_XRS + 258 * PHP_CRS;
    static int cus_dev[64] = {0};            // per device, idempotent -> thread-safe without a lock
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    int cus = __atomic_load_n(&cus_dev[dev & 63], __ATOMIC_ACQUIRE);
    if (cus == 0) {
        hipError_t e = hipFuncSetAttribute((const void*)hconv_pair_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        if ((e = hipFuncSetAttribute((const void*)hconv_pair_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)) != hipSuccess) return (int)e;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return (int)e;
        __atomic_store_n(&cus_dev[dev & 63], cus, __ATOMIC_RELEASE);
    }
    if (T <= 256) {      // one tile per pair
        hipLaunchKernelGGL(hconv_pair_kernel<false>, dim3(N < cus ? N : cus), dim3(512), LDS, st, (const _Float16*)x,
                           (const nef_h8*)wp1, b1, scale, (const nef_h8*)wp2, b2, (_Float16*)y, T, N, x_div, nq, sc_bs, sc_is, 1);
    } else {             // 252 output rows per tile
        const int tiles = (T + 251) / 252;
        const int64_t total = (int64_t)N * tiles;
        if (total > 0x7FFFFFFF) return NEF_E_SHAPE;
        hipLaunchKernelGGL(hconv_pair_kernel<true>, dim3((unsigned)(total < cus ? total : cus)), dim3(512), LDS, st, (const _Float16*)x,
                           (const nef_h8*)wp1, b1, scale, (const nef_h8*)wp2, b2, (_Float16*)y, T, N, x_div, nq, sc_bs, sc_is, tiles);
    }
    return nef_launch_status();
}

static int launch_hconv_tail(const void* x, const void* wp3, const float* b3, const void* wp4, const float* b4, const float* wout,
                             const float* bout, float* out, int N, int T, int nq, long out_bs, long out_is, hipStream_t st) {
    constexpr int LDS = 2 * 514 * PH_XRS + 3 * 512 * 4 + 8 * 64 * 16;      // two row buffers, d0 / d2 / d1, the last conv's A fragments
    static int cus_dev[64] = {0};            // per device, idempotent -> thread-safe without a lock
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    int cus = __atomic_load_n(&cus_dev[dev & 63], __ATOMIC_ACQUIRE);
    if (cus == 0) {
        hipError_t e = hipFuncSetAttribute((const void*)hconv_tail_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        if ((e = hipFuncSetAttribute((const void*)hconv_tail_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)) != hipSuccess) return (int)e;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return (int)e;
        __atomic_store_n(&cus_dev[dev & 63], cus, __ATOMIC_RELEASE);
    }
    if (T <= 512) {      // one tile per pair
        hipLaunchKernelGGL(hconv_tail_kernel<false>, dim3(N < cus ? N : cus), dim3(512), LDS, st, (const _Float16*)x, (const nef_h8*)wp3, b3,
                           (const nef_h8*)wp4, b4, wout, bout, out, T, N, nq, out_bs, out_is, 1);
    } else {             // 508 output rows per tile
        const int tiles = (T + 507) / 508;
        const int64_t total = (int64_t)N * tiles;
        if (total > 0x7FFFFFFF) return NEF_E_SHAPE;
        hipLaunchKernelGGL(hconv_tail_kernel<true>, dim3((unsigned)(total < cus ? total : cus)), dim3(512), LDS, st, (const _Float16*)x,
                           (const nef_h8*)wp3, b3, (const nef_h8*)wp4, b4, wout, bout, out, T, N, nq, out_bs, out_is, tiles);
    }
    return nef_launch_status();
}

static bool ph_l4_old() {
    // the 64 -> 64 layer (+ fused last conv) on hconv_wide_kernel<64, 64, 0, OUT> measured SLOWER than the round-1 kernel (sweep 52.1 vs
    // 50.7 ms, gen_ecg share 9.98 vs 9.68: one 64-channel chunk per tile leaves the double buffer nothing to hide): opt-in, A/B only
    static const bool v = !(nef_diag_env("NEF_PANO_L4_WIDE") && atoi(nef_diag_env("NEF_PANO_L4_WIDE")) == 1);
    return v;
}

extern "C" {

int nef_pano_h_from_f32(const float* x, void* y, int B, int C, int T, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(x && y, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && C > 0 && T > 0 && B <= 65535, NEF_E_SHAPE);
    hipLaunchKernelGGL(ph_transpose_kernel, dim3((T + 63) / 64, (C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, x,
                       (_Float16*)y, C, T);
    return nef_launch_status();
}

int nef_pano_h_pack_weight(const float* w, void* wp, int Cout, int Cin, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(w && wp, NEF_E_NULL);
    NEF_REQUIRE(Cout > 0 && Cin > 0 && Cout % 64 == 0 && Cin % 64 == 0, NEF_E_SHAPE);
    const int64_t total = (int64_t)Cout * Cin * 3;
    hipLaunchKernelGGL(ph_pack_weight_kernel, dim3(nef_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       (_Float16*)wp, Cout, Cin);
    return nef_launch_status();
}

int nef_pano_h_conv(const void* x, const void* wp, const float* bias, const float* scale, void* y, int N, int T, int Cin,
                    int Cout, int pro_mode, int x_div, int nq, int64_t sc_bs, int64_t sc_is, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(x && wp && bias && y, NEF_E_NULL);
    NEF_REQUIRE(N > 0 && T > 0 && x_div > 0 && nq > 0, NEF_E_SHAPE);
    NEF_REQUIRE(!(pro_mode & 1) || scale, NEF_E_NULL);
    NEF_REQUIRE(!(pro_mode & 2) || T % 2 == 0, NEF_E_SHAPE);
    hipStream_t st = (hipStream_t)stream;
#define PH_CASE(ci, co, pro) \
    if (Cin == ci && Cout == co && pro_mode == pro) \
        return launch_hconv<ci, co, pro>(x, wp, bias, scale, y, N, T, x_div, nq, sc_bs, sc_is, st)
#define PHW_CASE(ci, co, pro) \
    if (Cin == ci && Cout == co && pro_mode == pro) \
        return launch_hconv_wide<ci, co, pro>(x, wp, bias, scale, y, N, T, x_div, nq, sc_bs, sc_is, st)
    PHW_CASE(256, 128, 3);
    PHW_CASE(256, 128, 1);
    PHW_CASE(128, 128, 0);
    PHW_CASE(128, 64, 2);
    PHW_CASE(128, 64, 0);
    if (!ph_l4_old()) PHW_CASE(64, 64, 0);      // round 5, opt-in (NEF_DIAG=1 NEF_PANO_L4_WIDE=1): measured slower, see ph_l4_old()
#undef PHW_CASE
    PH_CASE(64, 64, 0);
#undef PH_CASE
    return NEF_E_UNSUPPORTED;
}

int nef_pano_h_conv_pair(const void* x, const void* wp1, const float* bias1, const float* scale, const void* wp2,
                         const float* bias2, void* y, int N, int T, int x_div, int nq, int64_t sc_bs, int64_t sc_is,
                         nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(x && wp1 && bias1 && scale && wp2 && bias2 && y, NEF_E_NULL);
    NEF_REQUIRE(N > 0 && T > 0 && T % 2 == 0 && x_div > 0 && nq > 0, NEF_E_SHAPE);
    return launch_hconv_pair(x, wp1, bias1, scale, wp2, bias2, y, N, T, x_div, nq, (long)sc_bs, (long)sc_is,
                             (hipStream_t)stream);
}

int nef_pano_h_conv_tail(const void* x, const void* wp3, const float* bias3, const void* wp4, const float* bias4, const float* wout,
                         const float* bout, float* out, int N, int T, int nq, int64_t out_bs, int64_t out_is, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(x && wp3 && bias3 && wp4 && bias4 && wout && bout && out, NEF_E_NULL);
    NEF_REQUIRE(N > 0 && T > 0 && T % 2 == 0 && nq > 0, NEF_E_SHAPE);
    return launch_hconv_tail(x, wp3, bias3, wp4, bias4, wout, bout, out, N, T, nq, (long)out_bs, (long)out_is, (hipStream_t)stream);
}

int nef_pano_h_conv_outconv(const void* x, const void* wp, const float* bias, const float* wout, const float* bout,
                            float* out, int N, int T, int nq, int64_t out_bs, int64_t out_is, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(x && wp && bias && wout && bout && out, NEF_E_NULL);
    NEF_REQUIRE(N > 0 && T > 0 && nq > 0, NEF_E_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    const int NT = 256;                      // tile width of the 64-channel layer
    const int tiles = (T + NT - 1) / NT;
    hipLaunchKernelGGL(ph_zero_edges_kernel, dim3(nef_stream_grid((int64_t)N * tiles * 2, 256)), dim3(256), 0, st, out, N,
                       T, NT, tiles, nq, (long)out_bs, (long)out_is);
    int rc = nef_launch_status();
    if (rc != NEF_OK) return rc;
    rc = ph_l4_old() ? launch_hconv<64, 64, 0, 1>(x, wp, bias, nullptr, nullptr, N, T, 1, nq, 0, 0, st, wout, out, (long)out_bs,
                                                  (long)out_is)
                     : launch_hconv_wide<64, 64, 0, 1>(x, wp, bias, nullptr, nullptr, N, T, 1, nq, 0, 0, st, wout, out,
                                                       (long)out_bs, (long)out_is);
    if (rc != NEF_OK) return rc;
    hipLaunchKernelGGL(ph_sigmoid3_kernel, dim3(nef_stream_grid((int64_t)N * T, 256)), dim3(256), 0, st, out, bout, N, T,
                       nq, (long)out_bs, (long)out_is);
    return nef_launch_status();
}

int nef_pano_h_outconv(const void* x, const float* w, const float* bias, float* out, int N, int T, int nq, int64_t out_bs,
                       int64_t out_is, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(x && w && bias && out, NEF_E_NULL);
    NEF_REQUIRE(N > 0 && T > 0 && nq > 0, NEF_E_SHAPE);
    const int tiles = (T + 255) / 256;
    hipLaunchKernelGGL(ph_outconv_kernel, dim3((unsigned)((int64_t)N * tiles)), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)x, w, bias, out, T, tiles, nq, (long)out_bs, (long)out_is);
    return nef_launch_status();
}

}  // extern "C"
