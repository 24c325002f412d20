/* nefnet_hip.h -- C ABI of libnefnet_hip.so: the gfx950 (MI355X) kernels of the Nef-Net train step.
 *
 * The reference (WhatAShot/Electrocardio-Panorama) has no FFI of its own: every op below replaces an
 * implicit PyTorch call site on the hot path, cited per entry as `codes/<file>:<line>`.  The Python
 * host (`electrocardio_panorama_amd/network`) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; the library never allocates;
 *   - activations are fp32 `[batch][channel][time]`, time contiguous; ROIs are int64 `[B][7][2]`;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises,
 *     every entry point is re-entrant and hipGraph-capturable; the only thing the library remembers is, per kernel and
 *     per device, that its dynamic-LDS limit has been raised (an idempotent, atomically published flag);
 *   - return 0 on success, a negative NEF_E_* for a rejected call, a positive value = hipError_t.
 */
#ifndef NEFNET_HIP_H
#define NEFNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NEF_OK 0
#define NEF_E_SHAPE (-1)        /* a dimension is out of the supported set */
#define NEF_E_NULL (-2)         /* a required pointer is NULL */
#define NEF_E_WORKSPACE (-3)    /* workspace too small */
#define NEF_E_UNSUPPORTED (-4)  /* configuration not built */

#define NEF_N_SEG 7    /* heartbeat segments per sample (codes/dataset/tianchi.py:103-106) */
#define NEF_ROI_BINS 16 /* roi_algin size (codes/network/model_nefnet.py:136) */

typedef void* nef_stream_t;

/* ABI version of this header; bumped on any signature change. */
int nef_abi_version(void);   /* 18 (round 6: nef_pack_desc + src_mode / src_Cr (polyphase weights synthesized inside the pack), + nef_amax_roll, nef_flatten, nef_regroup_halves, nef_step_words, nef_pano_h_conv_tail, + num_batches_tracked in the three BatchNorm statistics entry points, nef_pano_h_conv_pair for any length); 17 (round 5: + pro_mode 4 / 8 / 9, nef_poly_weights, nef_poly_fwd_edge, nef_poly_bwd_edge, nef_mix_bwd_shared: polyphase forward / backward-data through the x2 upsampling); 16 (round 5: + nef_set_option / nef_get_option); 15 (round 4: + x_clamped / clamped counters); 14 (round 4: + nef_conv_bwd_weight_h2); 13 (round 4: + nef_pack_weight_h2 / conv args wino = 3 and x_scale: direct convolutions on exact fp16 splits of the fp32 operands); 12 (round 4: nef_conv_bwd_weight_wino -- the transposed F(3,2) weight gradient -- is gone, nef_conv_bwd_weight_wino4 covers every shape it took; 11, round 3: the K = 7 F(4,.) operand has 13 planes, Winograd operands are laid out as 16-byte vectors; 10: + nef_pano_h_conv_pair) */

/* Kernel-form options of the process (tuning / A-B hooks; every value computes the same results).  Not part of any reference
 * interface: the reference's nn.Conv1d has one form (codes/network/model_nefnet.py:18-21).  Returns the previous value, or
 * NEF_E_SHAPE for an unknown key.  Read by the launches that follow; not meant to be flipped while a hipGraph capture is open. */
#define NEF_OPT_H2_FORM 1       /* split-fp16 forward / backward-data convolutions (conv args wino = 3): 0 (default) = every wave
                                   stages, multiplies and stores in turn (csrc/conv_h2.hip); 1 = producer and consumer waves in a
                                   persistent twelve-wave workgroup (tools/experiments/conv_h2p.hip; only in libraries built with
                                   `csrc/build.py --with-experiments`, ignored otherwise) wherever its shape rules hold -- bit-identical
                                   results, measured slower in round 5, kept as the A/B reference */
#define NEF_OPT_H2P_WGS 2       /* persistent workgroups per CU of that form (default 1: twelve waves fill a CU at 168 registers) */
int nef_set_option(int key, int value);
int nef_get_option(int key);

/* ---------------------------------------------------------------------------------------------
 * Stem: Conv1d(1->128 per lead, k15, s2, p7, no bias) + ReLU + MaxPool1d(3,2,1), fused.
 * Replaces codes/network/encoder/encoder.py:35-38 (resnet_1d.py:102-105).
 *   x [B][V][L], w [128V][1][15], y [B][128V][L/4]; L % 4 == 0.
 * bwd_weight recomputes the conv, routes gy through the pool arg-max and the ReLU.
 *   ws: nef_stem_bwd_ws_bytes(V) bytes of scratch. */
int nef_stem_fwd(const float* x, const float* w, float* y, int B, int V, int L, nef_stream_t stream);
size_t nef_stem_bwd_ws_bytes(int V);
int nef_stem_bwd_weight(const float* x, const float* w, const float* gy, float* gw, void* ws, size_t ws_bytes,
                        int B, int V, int L, nef_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Grouped Conv1d (stride 1, odd K in {1,3,7}, pad (K-1)/2) as an implicit GEMM on fp32 MFMA.
 * Replaces nn.Conv1d at codes/network/model_nefnet.py:18,21,32,44 and encoder/resnet_1d.py:23.
 *
 * nef_pack_weight: w [G*Cog][Cig][K] (torch layout) -> wp.
 *   transpose_flip = 0: forward operand   wp[g][k][ci][co] = w[g*Cog+co][ci][k]
 *   transpose_flip = 1: bwd-data operand  wp[g][k][co][ci] = w[g*Cog+co][ci][K-1-k]
 */
int nef_pack_weight(const float* w, float* wp, int G, int Cog, int Cig, int K, int transpose_flip,
                    nef_stream_t stream);

/* Winograd F(2,3) operand (transpose_flip = 1: the backward-data operand, (ci, co) exchanged and the taps reversed).
 * K == 3: wp[g][plane][ci][co], 4 planes (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2) of the taps w[g*Cog+co][ci][0..2].
 * K == 7 (taps split 4 + 3): 10 planes = the F(2,4) transform of taps 0..3 (g0/2, -(g0+g1+g2+g3)/2, (-g0+g1-g2+g3)/6,
 * (g0+2g1+4g2+8g3)/6, g3), the F(2,3) transform of taps 4..6 with its last plane negated, one plane of padding, laid out for
 * 16-byte fragment loads: wp[g][q][ci][co / 64][co % 32][4] with value 2*plane + (co % 64) / 32 = 4*q + e (conv_mfma.hip).
 * The layout is private to nef_conv_fwd; the size is planes * G * Cog * Cig floats either way. */
int nef_pack_weight_wino(const float* w, float* wp, int G, int Cog, int Cig, int K, int transpose_flip,
                         nef_stream_t stream);

/* The F(4,3) operand (conv args wino = 2): 6 planes for K == 3 -- (g0/4, -(g0+g1+g2)/6, -(g0-g1+g2)/6, g0/24+g1/12+g2/6,
 * g0/24-g1/12+g2/6, g2) -- and 13 for K == 7: taps split 4 + 3, the F(4,4) transform of taps 0..3 (7 planes, points 0, +-1,
 * +-2, inf, 1/2) followed by the F(4,3) transform of taps 4..6.  Stored as slabs of 16-byte vectors,
 * wp[g][plane / 4][ci][co / 32][co % 32][4] plus a tail slab for the planes % 4 last planes (conv_mfma.hip); planes * G *
 * Cog * Cig floats. */
int nef_pack_weight_wino4(const float* w, float* wp, int G, int Cog, int Cig, int K, int transpose_flip,
                          nef_stream_t stream);

/* The split-fp16 operand (conv args wino = 3): each weight, scaled by a power of two per output row of the launch (row
 * maximum -> [2^14, 2^15)), is split exactly into two fp16 terms w = wh + wl (+ < 2^-22 |w|) and stored as matrix-core
 * fragments wp[g][ci/16][k][co/32][wh | wl][lane][8] (fp16), followed by the per-row descale factors [g][co] (fp32).
 * nef_conv_fwd then runs the DIRECT convolution as three v_mfma_f32_32x32x16_f16 per 16 channels and tap -- xh*wh + xh*wl +
 * xl*wh, exact fp16 products, fp32 accumulation (csrc/conv_h2.hip): fp32-class results (1.1e-7 rel-L2 on a 128-channel K = 7
 * layer against fp64; an fp32 direct conv: 1.3e-7) at 3/16 of the fp32 matrix instructions' pipe time.  K == 3 or 7, Cig % 16 == 0,
 * Cog % 32 == 0 (both as the LAUNCH sees them: exchanged under transpose_flip).  wp: nef_pack_weight_h2_bytes(...) bytes. */
int nef_pack_weight_h2(const float* w, void* wp, int G, int Cog, int Cig, int K, int transpose_flip, nef_stream_t stream);
size_t nef_pack_weight_h2_bytes(int G, int Cog, int Cig, int K, int transpose_flip);

/* Any number of operands in one launch (a whole forward or backward pass packs ~25): each descriptor is one
 * nef_pack_weight (wino = 0), nef_pack_weight_wino (wino = 1), nef_pack_weight_wino4 (wino = 2) or nef_pack_weight_h2 (wino = 3) call.  `descs` is a HOST array. */
typedef struct nef_pack_desc {
    const float* w;
    float* wp;
    int32_t G, Cog, Cig, K, transpose_flip, wino;
    /* round 6, wino = 3 only: how the packed tensor [G*Cog][Cig][K] is read out of `w`.  0: `w` is that tensor.  1: `w` is
     * [G*Cog/2][Cig][3] and the packed tensor holds the two PHASE weights of each row (conv1d(upsample2(x), w) as two K = 3 convs on the
     * half-resolution x, codes/network/model_nefnet.py:102-105 -- what nef_poly_weights would write, formed inside the pack so the
     * phase tensor is never materialised); src_Cr = 0: row 2 r + p, src_Cr = Cog/2 > 0: the tile order of the polyphase forward launch. */
    int32_t src_mode, src_Cr;
} nef_pack_desc;
int nef_pack_weights(const nef_pack_desc* descs, int n, nef_stream_t stream);

typedef struct nef_conv_args {
    const float* x;        /* input  [B][..][T]; element (b, g, ci, t) at x + b*x_bs + g*x_gs + ci*T + t */
    const float* wp;       /* packed weights [G][K][Cin_g][Cout_g] (nef_pack_weight) */
    float* y;              /* output; element (b, g, co, t) at y + b*y_bs + g*y_gs + co*T + t */
    const float* bias;     /* [G*Cout_g] or NULL */
    const float* in_scale; /* NULL, or per (sample, input channel) factor at in_scale + b*sc_bs + g*sc_gs + ci */
    const float* res;      /* NULL, or residual added before the activation, laid out like y with res_bs/res_gs */
    const float* gate;     /* NULL, or y *= gate_scale * (gate > 0), laid out like y with gate_bs/gate_gs */
    const uint8_t* mask;   /* NULL, or dropout keep-mask (0/1), dense [B][G*Cout_g][T]; y *= mask * drop_scale */
    int64_t x_bs, x_gs, y_bs, y_gs, sc_bs, sc_gs, res_bs, res_gs, gate_bs, gate_gs;
    int32_t B, T, G, Cin_g, Cout_g, K;
    int32_t relu;          /* apply max(0, .) after bias + residual */
    float gate_scale;
    float drop_scale;      /* 1/(1-p); with mask == NULL and drop_p > 0 the keep-mask comes from the counter RNG */
    float drop_p;
    uint64_t rng_seed;     /* counter RNG: keep(b, channel, t) = hash(seed, dense index) >= p */
    /* Input prologue, applied while the activation tile is staged (decoder fusion, K == 3 only):
     *   pro_mode bit0: x' = max(0, x*pro_a[p][ch] + pro_b[p][ch]) with p = sample / pro_Bp -- the BatchNorm affine + ReLU
     *                  of the producing layer (model_nefnet.py:19-23); pro_a/pro_b are [P][G*Cin_g];
     *   pro_mode bit1: the input is stored at half resolution [..][T/2] and is x2-upsampled on the fly exactly as
     *                  nn.Upsample(scale_factor=2, mode='linear', align_corners=False) (model_nefnet.py:102,104);
     *                  x_bs / x_gs and the channel pitch then refer to the half-resolution tensor.
     *   pro_mode 8 / 9 (wino == 3 only): polyphase forward of conv1d(upsample2(x)) -- x at half resolution [..][T], Cout_g = 2 x the
     *                  conv's channels (tile-ordered phase weights, nef_poly_weights), y [..][Cout_g / 2][2 T]; bit0 as above;
     *                  bias and stats only; followed by nef_poly_fwd_edge.
     *   pro_mode 4 (alone, wino == 3 only): phase-stacked input for the polyphase backward-data pass -- x is a full-resolution
     *                  tensor [..][Cin_g / 2][2 T]; reduction channel 2 c + p at position m is x[c][2 m + p] (see nef_poly_weights).
     * Zero padding is applied after the prologue.  in_scale must be NULL when pro_mode != 0. */
    const float* pro_a;
    const float* pro_b;
    int32_t pro_mode;
    int32_t pro_Bp;
    const uint64_t* rng_seed_dev;  /* NULL, or a device word added to rng_seed at run time (hipGraph replay: a captured
                                      launch freezes its arguments, the per-step seed must live in device memory) */
    int32_t wino;          /* 1: wp was packed by nef_pack_weight_wino -- K == 3 through Winograd F(2,3), K == 7 through
                              F(2,4) + F(2,3) (2/3 resp. 9/14 of the multiplies; still fp32 multiplies and adds on the matrix cores, results differ
                              from the direct form by the rounding of the transforms).  2: packed by
                              nef_pack_weight_wino4 -- Winograd F(4,3) resp. F(4,4) + F(4,3): 1/2 resp. 13/28 of the multiplies.  Needs T even, T >= 128
                              (Cout_g % 128 == 0) or T >= 256 (Cout_g % 64 == 0), Cin_g % 16 == 0; K == 7: pro_mode 0.
                              3: packed by nef_pack_weight_h2 -- the direct convolution on exact fp16 splits of both operands
                              (K == 1, 3 or 7, Cout_g % 64 == 0, Cin_g % 16 == 0, T even and >= 128; every epilogue option
                              incl. stats / bnb_slots; K == 7: pro_mode 0).  Short rows, 8 <= T <= 64 with T % 4 == 0
                              (K == 1 or 3): several samples per tile; then no prologue, in_scale, stats or bnb_slots. */
    float* stats;          /* NULL, or (wino == 2 only) the epilogue also leaves, per output channel and per 128-column slot
                              of a sample, the sum and the sum of squares of the final outputs of that slot:
                              stats[(ch * B * nslot + b * nslot + slot) * 2 + {0,1}], ch = g*Cout_g + co, nslot =
                              nef_conv_stats_slots(T, Cout_g) -- the train-mode BatchNorm statistics of the conv output
                              (model_nefnet.py:19,22) without a second pass over it; finished by nef_bn_stats_from_slots */
    /* BatchNorm-BACKWARD sums of the layer this (backward-data) launch propagates into, wino == 2 only, not together with
     * `stats`: with bnb_slots != NULL the epilogue also leaves, in the same slot layout, sum(g*m) and sum(g*m*xhat) of its
     * final outputs g, where m = [bnb_x*bnb_a + bnb_b > 0] and xhat = (bnb_x - bnb_mean)*bnb_invstd -- the reduction pass
     * of nef_bn_relu_bwd over (g, x), which then takes `slots` instead.  bnb_x: the BatchNorm input, dense
     * [B][G*Cout_g][T]; bnb_mean/invstd/a/b: [P][G*Cout_g], pass p = sample / bnb_Bp. */
    const float* bnb_x;
    const float* bnb_mean;
    const float* bnb_invstd;
    const float* bnb_a;
    const float* bnb_b;
    float* bnb_slots;
    int32_t bnb_Bp;
    int32_t bnb_up;        /* 1: a x2 linear upsampling (nn.Upsample, align_corners=False) sits between that BatchNorm's ReLU
                              and this launch's output: bnb_x is [B][G*Cout_g][T/2] and the sums are those of the
                              upsampling's adjoint (what nef_bn_relu_bwd_up reduces); T % 4 == 0 */
    float x_scale;         /* wino == 3 only: 0 (= 1) or an exact power of two the input is multiplied by before it is split into
                              fp16 terms and the accumulators are divided by again -- brings operands far from magnitude 1
                              (gradients) into fp16's range; the result is unchanged up to the split's rounding */
    int32_t reserved0;
    const float* x_amax;   /* wino == 3: NULL, or a device word holding the largest |input| (after in_scale / prologue) this call
                              site saw at its previous launch: when it is a positive finite number the input scale is derived
                              from it (-> [2^8, 2^9)) instead of x_scale -- the operand is then in fp16's range whatever its
                              magnitude, and the launch stays capturable (nothing is read back by the host) */
    float* x_amax_next;    /* wino == 3: NULL, or a device word (zeroed by the caller) this launch max-accumulates the largest
                              |input| of ITS operand into -- next launch's x_amax */
    int32_t* x_clamped;    /* wino == 3: NULL, or a device counter the launch adds 1 to (per wave) when an element of its operand is
                              still out of fp16's range AFTER the range rescue -- i.e. is not finite.  (A tile whose finite data does
                              not fit under the launch's scale -- the operand grew more than ~64x since x_amax was measured -- is
                              redone inside the launch with the scale its own data asks for; round 4 clamped it and counted it.)
                              The producer / consumer form (nef_set_option(NEF_OPT_H2_FORM, 1)) has no rescue: it clamps at
                              65000 / scale and counts */
    const float* res_scale;   /* wino == 3, res != NULL: NULL, or a per-(sample, channel) factor on the residual,
                              y = conv + bias + res * res_scale[b*rs_bs + g*rs_gs + c] -- the residual of a block whose input is a
                              channel-scaled tensor (w_conv behind the theta scaling, codes/network/model_nefnet.py:122-124) read
                              from the UNSCALED tensor, with in_scale on the block's first conv: the scaled tensor is never written */
    int64_t rs_bs, rs_gs;
    const float* gate_rowscale;   /* wino == 3, gate != NULL: NULL, or a per-(sample, channel) factor on the gated output,
                              y = gate > 0 ? y * gate_scale * gate_rowscale[b*gr_bs + g*gr_gs + c] : 0 -- the backward of the same
                              channel scaling (nef_chscale_bwd's gx) taken in the epilogue of the block's last backward-data launch */
    int64_t gr_bs, gr_gs;
    int32_t stats_mode;    /* what `stats` receives: 0 = per-slot sum and sum of squares of the outputs; 1 (wino == 3, gate != NULL) =
                              per-slot sum of (ungated, unscaled output) x gate in word 0, 0 in word 1 -- nef_chscale_bwd's gs[b][c]
                              = sum_t gy x, finished by nef_slots_to_rows */
    int32_t reserved1;
} nef_conv_args;

/* y = epilogue(conv(x * in_scale, wp) + bias + res).  Also the bwd-data pass (pack with transpose_flip=1,
 * swap Cin_g/Cout_g). */
int nef_conv_fwd(const nef_conv_args* a, nef_stream_t stream);
/* sizeof(nef_conv_args) as the library was built: a binding checks its mirror of the struct against it. */
size_t nef_conv_args_bytes(void);

/* gw[g*Cog+co][ci][k] = sum_{b,t} gy[b][g][co][t] * (x*in_scale)[b][g][ci][t+k-pad].  gw is overwritten.
 * ws: nef_conv_bwd_weight_ws_bytes(...) bytes of scratch (split-K partials, reduced deterministically). */
size_t nef_conv_bwd_weight_ws_bytes(int B, int T, int G, int Cin_g, int Cout_g, int K);
int nef_conv_bwd_weight(const float* x, int64_t x_bs, int64_t x_gs, const float* in_scale, int64_t sc_bs,
                        int64_t sc_gs, const float* gy, int64_t gy_bs, int64_t gy_gs, float* gw, void* ws,
                        size_t ws_bytes, int B, int T, int G, int Cin_g, int Cout_g, int K, nef_stream_t stream);
/* Same with the input prologue of nef_conv_args (pro_mode / pro_a / pro_b / pro_Bp) recomputed while staging x. */
int nef_conv_bwd_weight_pro(const float* x, int64_t x_bs, int64_t x_gs, const float* pro_a, const float* pro_b,
                            int pro_mode, int pro_Bp, const float* gy, int64_t gy_bs, int64_t gy_gs, float* gw, void* ws,
                            size_t ws_bytes, int B, int T, int G, int Cin_g, int Cout_g, int K, nef_stream_t stream);

/* The same weight gradient (either form above: in_scale or the input prologue) through the TRANSPOSED Winograd algorithm;
 * fp32 multiplies and adds on the matrix cores, results differ from the direct form by the rounding of the transforms.
 * Needs T even and T >= 64; K == 7: pro_mode 0.  Workspace as for nef_conv_bwd_weight.
 *   K == 3: transposed F(3,4), 6 multiplies per four columns -- 1/2 of the direct form's; transform entries up to 8 and 1/24
 *           (the F(4,3) matrices of the forward kernels, roles exchanged): measured rounding 1..6x the direct form's.
 *   K == 7: the taps split 4 + 3 across TWO launches, transposed F(4,4) + transposed F(3,4): 13 multiplies per 8 columns
 *           (direct: 28).
 * Which kernel runs: the (gy, x) tiles are streamed by LDS-DMA through a ring of LDS buffers (csrc/conv_bww_glds.hip) when
 * in_scale == NULL, both channel counts are multiples of 64, T >= 64, T % 4 == 0 if pro_mode has the upsampling bit, not
 * both prologue bits at once, and at most 8 BatchNorm passes (B / pro_Bp); otherwise the register-staged kernel of
 * csrc/conv_mfma.hip.  Same arithmetic and partial-sum layout; the two differ only by the summation order across splits.
 * The kernel never reads outside [x, x + (B-1)*x_bs + (G-1)*x_gs + Cin_g*T) resp. the same extent of gy.
 * (ABI <= 11 also had nef_conv_bwd_weight_wino, the transposed F(3,2): removed, this entry covers every shape it took.) */
int nef_conv_bwd_weight_wino4(const float* x, int64_t x_bs, int64_t x_gs, const float* in_scale, int64_t sc_bs,
                              int64_t sc_gs, const float* pro_a, const float* pro_b, int pro_mode, int pro_Bp,
                              const float* gy, int64_t gy_bs, int64_t gy_gs, float* gw, void* ws, size_t ws_bytes, int B, int T,
                              int G, int Cin_g, int Cout_g, int K, nef_stream_t stream);

/* The same weight gradient on exact fp16 splits of BOTH operands (csrc/conv_h2w.hip; the arithmetic of nef_pack_weight_h2 /
 * conv args wino = 3: gy = gh + gl, X = xh + xl, three fp16 matrix instructions per product, fp32 accumulation): fp32-class
 * results (closer to fp64 than the transposed-Winograd forms) at 3/16 of the fp32 matrix instructions' pipe time.  K == 3 (any
 * pro_mode) or K == 7 (pro_mode 0), T even and >= 64, both channel counts multiples of 64; in_scale only with pro_mode 0.
 * x_scale / gy_scale: 0 (= 1) or the exact powers of two the operands are multiplied by before they are split (their product is
 * divided out); x_amax / gy_amax: NULL, or device words with the largest |operand| the call site saw before -- when positive and
 * finite the scales are derived from them instead; x_amax_next / gy_amax_next: both NULL, or device words this launch
 * max-accumulates its operands' magnitudes into; clamped: NULL, or a device counter incremented when a scaled operand element
 * had to be clamped at fp16's range (see nef_conv_args.x_clamped).  ws: nef_conv_bwd_weight_h2_ws_bytes(...) bytes (split partial sums, added up in a
 * fixed order). */
size_t nef_conv_bwd_weight_h2_ws_bytes(int B, int T, int G, int Cin_g, int Cout_g, int K);
int nef_conv_bwd_weight_h2(const float* x, int64_t x_bs, int64_t x_gs, const float* in_scale, int64_t sc_bs, int64_t sc_gs,
                           const float* pro_a, const float* pro_b, int pro_mode, int pro_Bp, const float* gy, int64_t gy_bs,
                           int64_t gy_gs, float* gw, void* ws, size_t ws_bytes, int B, int T, int G, int Cin_g, int Cout_g, int K,
                           float x_scale, float gy_scale, const float* x_amax, const float* gy_amax, float* x_amax_next,
                           float* gy_amax_next, int32_t* clamped, nef_stream_t stream);

/* out[c] = sum_{b,t} x[b][c][t] (bias gradients).  ws: nef_chan_sum_ws_bytes(C). */
size_t nef_chan_sum_ws_bytes(int C);
int nef_chan_sum(const float* x, float* out, void* ws, size_t ws_bytes, int B, int C, int T, nef_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * ConvTranspose1d(k=2, s=2, groups=G, 128 -> 64 per group, bias).  codes/network/model_nefnet.py:96-97.
 *   x [B][G*Cig][T], w [G*Cig][Cog][2], bias [G*Cog], y [B][G*Cog][2T]. */
int nef_convt2_fwd(const float* x, const float* w, const float* bias, float* y, int B, int G, int Cig, int Cog,
                   int T, nef_stream_t stream);
int nef_convt2_bwd_data(const float* gy, const float* w, float* gx, int B, int G, int Cig, int Cog, int T,
                        nef_stream_t stream);
/* The same op through the matrix cores: a grouped 1x1 conv onto m = co*2+j (nef_conv_fwd, K=1, weights
 * nef_group_transpose'd to [G][2Cog][Cig]) + these layout passes:
 *   nef_group_transpose    : out[g][c][r] = in[g][r][c]                     (weights both ways: the op is an involution)
 *   nef_convt2_interleave  : y[b][c][2t+j] = yq[b][2c+j][t] + bias[c]       (bias may be NULL), C = G*Cog
 *   nef_convt2_deinterleave: gyq[b][2c+j][t] = gy[b][c][2t+j] */
int nef_group_transpose(const float* in, float* out, int G, int R, int Cn, nef_stream_t stream);
int nef_convt2_interleave(const float* yq, const float* bias, float* y, int B, int C, int T, nef_stream_t stream);
int nef_convt2_deinterleave(const float* gy, float* gyq, int B, int C, int T, nef_stream_t stream);
size_t nef_convt2_bwd_weight_ws_bytes(int G, int Cig, int Cog);
int nef_convt2_bwd_weight(const float* x, const float* gy, float* gw, float* gb, void* ws, size_t ws_bytes, int B,
                          int G, int Cig, int Cog, int T, nef_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Angular encoding + Linear(12 -> O).  codes/network/utils/theta_encoder.py:13-29 with
 * model_nefnet.py:76-77,121,164,183.   theta [N][2], W [O][12], bias [O], y [N][O]. */
int nef_theta_mlp_fwd(const float* theta, const float* W, const float* bias, float* y, int N, int O,
                      nef_stream_t stream);
int nef_theta_mlp_bwd(const float* theta, const float* gy, float* gW, float* gb, int N, int O, nef_stream_t stream);
/* enc [N][12] only (test hook for the encoding itself). */
int nef_theta_encode(const float* theta, float* enc, int N, nef_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Per-(sample, channel) scaling.  model_nefnet.py:122-123 (lead latents x mlp1) and :166,170,174,186.
 *   y[b][c][t] = x[b][c][t] * s[b*s_bs + c];  bwd: gx = gy * s, gs[b][c] = sum_t gy * x (gs dense [B][C]). */
int nef_chscale_fwd(const float* x, const float* s, int64_t s_bs, float* y, int B, int C, int T, nef_stream_t stream);
/* relu_x != 0: x is a ReLU output; gx is additionally masked with x > 0 (the gate its producer's backward starts with). */
int nef_chscale_bwd(const float* gy, const float* x, const float* s, int64_t s_bs, float* gx, float* gs, int B,
                    int C, int T, int relu_x, nef_stream_t stream);

/* out = g * scale * (ref > 0)   (ReLU / dropout back-propagation gate), n elements. */
int nef_gate(const float* g, const float* ref, float* out, float scale, int64_t n, nef_stream_t stream);
/* out = a + b, n elements. */
int nef_add(const float* a, const float* b, float* out, int64_t n, nef_stream_t stream);
/* Time-window copy of a grouped view: dst [B][G*Cg][W] dense <- src(b,g,c, t0 + w)  (src strides x_bs / x_gs, rows T). */
int nef_window_crop(const float* src, int64_t x_bs, int64_t x_gs, float* dst, int B, int G, int Cg, int T, int t0, int W,
                    nef_stream_t stream);
/* Inverse: dst(b,g,c,t) = (t0 <= t < t0+W) ? src[b][g*Cg+c][t-t0] : 0 over the whole grouped view (zero fill). */
int nef_window_scatter(const float* src, float* dst, int64_t y_bs, int64_t y_gs, int B, int G, int Cg, int T, int t0,
                       int W, nef_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * ROI ops.  rois: int64 [B][7][2] in input-sample units; latent index = roi * 0.25 (model_nefnet.py:136,143).
 * nef_roi_align_*: codes/network/utils/roi_pooling_1d.py:38-69 with the reference's actual semantics
 *   (grid x addresses the size-1 axis).  z [B][C][T] -> out [B][C][7][16].  Only the two middle rows of
 *   time are ever read, so z / gz may store just a window: zT samples per row starting at time t_off (zT=T, t_off=0
 *   for a full tensor); the window must contain rows (T-1)/2 and (T-1)/2+1.
 * nef_roi_unpool_*: roi_pooling_1d.py:72-99.  zseg [B][C][7][32] -> out [B][C][T].
 *   status (int32[1], may be NULL): set to 1 if a sample's segment lengths are negative or do not sum to T. */
int nef_roi_align_fwd(const float* z, const int64_t* rois, float* out, int B, int C, int T, int zT, int t_off,
                      nef_stream_t stream);
int nef_roi_align_bwd(const float* gout, const int64_t* rois, float* gz, int B, int C, int T, int zT, int t_off,
                      nef_stream_t stream);
int nef_roi_unpool_fwd(const float* zseg, const int64_t* rois, float* out, int32_t* status, int B, int C, int T,
                       nef_stream_t stream);
int nef_roi_unpool_bwd(const float* gout, const int64_t* rois, float* gzseg, int B, int C, int T,
                       nef_stream_t stream);
/* seg_start/seg_len int64 [B][7]: the integer bookkeeping of roi_pooling_1d.py:82-92 (bit-exact test hook). */
int nef_roi_segment_table(const int64_t* rois, int64_t* seg_start, int64_t* seg_len, int B, nef_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Lead mean / Standin shuffle / query scaling.  model_nefnet.py:146-176.
 *   z1, z2r [B][128V][T]; latent [B][256][T] = cat(mean_v z1, mean_v z2r);
 *   q [B][256];  D [3][B][256][T]:  D0 = q*latent, D1 = q*cat(z1[c1], z2mean), D2 = q*cat(z1mean, z2r[c2]).
 *   choice_dev: NULL, or device int32[2] = {c1, c2} that overrides the arguments (hipGraph replay). */
int nef_lead_mean(const float* z1, const float* z2r, float* latent, int B, int V, int T, nef_stream_t stream);
int nef_mix_fwd(const float* latent, const float* z1, const float* z2r, const float* q, float* D, int B, int V,
                int T, int c1, int c2, const int32_t* choice_dev, nef_stream_t stream);
int nef_mix_bwd(const float* gD, const float* latent, const float* z1, const float* z2r, const float* q, float* gz1,
                float* gz2r, float* gq, int B, int V, int T, int c1, int c2, const int32_t* choice_dev, int relu_z1,
                nef_stream_t stream);
/* relu_z1 != 0 (both): z1 is a ReLU output; gz1 is additionally masked with z1 > 0.
 * nef_mix_bwd_up: same, but gU is the gradient wrt the x2-UPSAMPLED decoder input [3B][256][2T] (what the first decoder
 * conv's backward-data writes); the upsampling adjoint (nef_upsample2_bwd) is taken while reading it. */
int nef_mix_bwd_up(const float* gU, const float* latent, const float* z1, const float* z2r, const float* q, float* gz1,
                float* gz2r, float* gq, int B, int V, int T, int c1, int c2, const int32_t* choice_dev, int relu_z1,
                nef_stream_t stream);

/* Shared first decoder conv (train step): the three decoder inputs of model_nefnet.py:159-176 are
 * q*cat(z1m|z2m), q*cat(z1[c1]|z2m), q*cat(z1m|z2r[c2]) and decoder.1.double_conv.0 is linear in the two channel halves,
 * so each distinct half goes through its half of the conv once (4 half-convs instead of 6):
 *   nef_mix_fwd_shared   : D2 [2B][256][T] = (q*cat(z1m|z2m) | q*cat(z1[c1]|z2r[c2]))
 *   (grouped conv, G = 2, on D2 with the weight regrouped to [2*Cout][Cin/2][3] -> P2 [2B][2C][L])
 *   nef_pass_combine_fwd : c1[p][b][c] = P2 A-half[ia(p)] + P2 B-half[ib(p)] + bias[c],  (ia,ib) = (m,m),(pick,m),(m,pick)
 *   nef_pass_combine_bwd : its adjoint, gc1 [3B][C][L] -> gP2 [2B][2C][L]
 *   nef_mix_bwd_shared_up: nef_mix_bwd_up for the two-pass gradient gU2 [2B][256][2T] (wrt the upsampled D2);
 *   nef_mix_bwd_shared: the same for the gradient wrt D2 itself, gD2 [2B][256][T] (what the polyphase backward-data pass leaves). */
int nef_mix_fwd_shared(const float* latent, const float* z1, const float* z2r, const float* q, float* D2, int B, int V,
                       int T, int c1, int c2, const int32_t* choice_dev, nef_stream_t stream);
/* nef_lead_mean + nef_mix_fwd_shared in one pass over z1 / z2r (the picked lead is one of the rows being averaged):
 * bit-identical latent and D2, 0.65 GB less traffic per step at config 2. */
int nef_lead_mean_mix_shared(const float* z1, const float* z2r, const float* q, float* latent, float* D2, int B, int V,
                             int T, int c1, int c2, const int32_t* choice_dev, nef_stream_t stream);
int nef_mix_bwd_shared(const float* gD2, const float* latent, const float* z1, const float* z2r, const float* q,
                       float* gz1, float* gz2r, float* gq, int B, int V, int T, int c1, int c2,
                       const int32_t* choice_dev, int relu_z1, nef_stream_t stream);
int nef_mix_bwd_shared_up(const float* gU2, const float* latent, const float* z1, const float* z2r, const float* q,
                          float* gz1, float* gz2r, float* gq, int B, int V, int T, int c1, int c2,
                          const int32_t* choice_dev, int relu_z1, nef_stream_t stream);
int nef_pass_combine_fwd(const float* P2, const float* bias, float* c1, int B, int C, int L, nef_stream_t stream);
/* nef_pass_combine_fwd that also leaves the train-mode BatchNorm statistics of its output (3 passes of B samples; same
 * outputs as nef_bn_train_stats(c1, ..., P = 3, Bp = B, ...), running statistics updated pass by pass): saves the separate
 * statistics pass over c1.  ws: nef_pass_combine_stats_ws_bytes(B, C). */
size_t nef_pass_combine_stats_ws_bytes(int B, int C);
int nef_pass_combine_fwd_stats(const float* P2, const float* bias, float* c1, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float* mean, float* invstd, float* a, float* b,
                               void* ws, size_t ws_bytes, int B, int C, int L, float eps, float momentum,
                               int64_t* num_batches_tracked, nef_stream_t stream);
int nef_pass_combine_bwd(const float* gc1, float* gP2, int B, int C, int L, nef_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Decoder pieces.  model_nefnet.py:10-27,101-107.
 * Upsample(scale 2, linear, align_corners=False): x [N][Tin] rows -> y [N][2Tin]. */
int nef_upsample2_fwd(const float* x, float* y, int64_t N, int Tin, nef_stream_t stream);
int nef_upsample2_bwd(const float* gy, float* gx, int64_t N, int Tin, nef_stream_t stream);
/* y [N][C][2Tin] = upsample2(max(0, x*a[p][c] + b[p][c])), p = n / Bp: BN affine + ReLU folded into the resample. */
int nef_upsample2_aff_fwd(const float* x, const float* a, const float* b, float* y, int N, int C, int Tin, int Bp,
                          nef_stream_t stream);

/* BatchNorm1d, training mode, P independent passes stacked along batch (x [P*Bp][C][L]).
 * nef_bn_train_stats: per (pass, channel) batch mean / biased var -> mean, invstd [P][C], the affine
 *   a = gamma*invstd, b = beta - mean*a [P][C]; then updates running_mean/var sequentially over passes with
 *   momentum (unbiased var), exactly as P successive module calls would.
 * nef_bn_eval_affine: a = gamma/sqrt(rv+eps), b = beta - rm*a  [C].
 * nef_affine_relu_fwd: y = max(0, x*a[p][c] + b[p][c]).
 * nef_bn_relu_bwd: given gy (grad wrt the ReLU output), x (pre-BN), writes gx, ggamma[C], gbeta[C] and, when
 *   gx_chan_sum != NULL, gx_chan_sum[c] = sum_{b,t} gx (the bias gradient of the conv that feeds this BN), fused into
 *   the same pass.  ws: nef_bn_bwd_ws_bytes(P, Bp, C). */
size_t nef_bn_ws_bytes(int P, int C);
/* Slots per sample of nef_conv_args.stats (0: the F(4,3) kernel does not take this shape), and the statistics pass that
 * replaces nef_bn_train_stats when the producing conv left them: same outputs (fp64 from the slot sums on, fixed
 * order), x is not read.  slots: [C][P*Bp*nslot][2] floats.  ws: nef_bn_ws_bytes(P, C). */
int nef_conv_stats_slots(int T, int Cout_g);
int nef_bn_stats_from_slots(const float* slots, int nslot, const float* gamma, const float* beta, float* running_mean,
                            float* running_var, float* mean, float* invstd, float* a, float* b, void* ws,
                            size_t ws_bytes, int P, int Bp, int C, int L, float eps, float momentum,
                            int64_t* num_batches_tracked, nef_stream_t stream);
int nef_bn_train_stats(const float* x, const float* gamma, const float* beta, float* running_mean,
                       float* running_var, float* mean, float* invstd, float* a, float* b, void* ws, size_t ws_bytes,
                       int P, int Bp, int C, int L, float eps, float momentum, int64_t* num_batches_tracked,
                       nef_stream_t stream);
/* (round 6) num_batches_tracked: NULL, or the BatchNorm's int64 counter, incremented by the number of passes P by the same launch
 * that updates the running statistics (nn.BatchNorm1d in train mode, codes/network/model_nefnet.py:19,22) -- in all three entry
 * points above (nef_pass_combine_fwd_stats: P = 3). */
int nef_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float* a, float* b, int C, float eps, nef_stream_t stream);
/* Eval-mode BN folded into the preceding conv (inference sweep): w_out[co][:] = a[co]*w[co][:], bias_out = a*bias + b;
 * w [Cout][inner]. */
int nef_fold_bn(const float* w, const float* bias, const float* a, const float* b, float* w_out, float* bias_out,
                int Cout, int inner, nef_stream_t stream);
int nef_affine_relu_fwd(const float* x, const float* a, const float* b, float* y, int P, int Bp, int C, int L,
                        nef_stream_t stream);
size_t nef_bn_bwd_ws_bytes(int P, int Bp, int C);
int nef_bn_relu_bwd(const float* gy, const float* x, const float* gamma, const float* mean, const float* invstd,
                    const float* a, const float* b, float* gx, float* ggamma, float* gbeta, float* gx_chan_sum, void* ws,
                    size_t ws_bytes, int P, int Bp, int C, int L, const float* slots, int nslot, nef_stream_t stream);
/* `slots` (here and in nef_bn_relu_bwd_combine3): NULL, or the sums the conv that produced gy left in its epilogue
 * (nef_conv_args.bnb_slots, nslot = nef_conv_stats_slots): the reduction pass over (gy, x) is then skipped. */

/* Final Conv1d(64->1,k3,p1,bias) + sigmoid(x/3).  model_nefnet.py:106,168.
 *   x [N][C][L], w [1][C][3], bias [1], out [N][L]. */
/* BatchNorm+ReLU backward of the LAST decoder BatchNorm fed straight from the last conv's output gradient: the
 * [N][C][L] input gradient of Conv1d(C->1) (what nef_outconv_bwd_data would write) is rebuilt on the fly from
 * go = gout*out*(1-out)/3, so it is never materialised.  Same results as nef_outconv_bwd_data + nef_bn_relu_bwd.
 * L % 4 == 0.  wout [1][C][3].  ws: nef_bn_bwd_outconv_ws_bytes(P, Bp, C, L). */
/* nef_bn_relu_bwd_up: nef_bn_relu_bwd whose incoming gradient is given at TWICE the length, gu [N][C][2L] = the
 * gradient wrt nn.Upsample(x2)(ReLU(BN(x))) (model_nefnet.py:104); the upsampling adjoint (nef_upsample2_bwd) is taken
 * while reading.  L % 4 == 0, L >= 8.  ws: nef_bn_bwd_ws_bytes. */
int nef_bn_relu_bwd_up(const float* gu, const float* x, const float* mean, const float* invstd, const float* a,
                       const float* b, float* gx, float* ggamma, float* gbeta, float* gx_chan_sum, void* ws,
                       size_t ws_bytes, int P, int Bp, int C, int L, const float* slots, int nslot, nef_stream_t stream);
/* nef_bn_relu_bwd_combine3: nef_bn_relu_bwd (P = 3) followed by nef_pass_combine_bwd in one pass: writes gP2 [2Bp][2C][L]
 * instead of gx (only gx's per-channel sum is kept).  ws: nef_bn_bwd_ws_bytes(3, Bp, C). */
int nef_bn_relu_bwd_combine3(const float* gy, const float* x, const float* mean, const float* invstd, const float* a,
                             const float* b, float* gP2, float* ggamma, float* gbeta, float* gx_chan_sum, void* ws,
                             size_t ws_bytes, int Bp, int C, int L, const float* slots, int nslot, nef_stream_t stream);
size_t nef_bn_bwd_outconv_ws_bytes(int P, int Bp, int C, int L);
int nef_bn_relu_bwd_outconv(const float* gout, const float* out, const float* wout, const float* x, const float* mean,
                            const float* invstd, const float* a, const float* b, float* gx, float* ggamma, float* gbeta,
                            float* gx_chan_sum, void* ws, size_t ws_bytes, int P, int Bp, int C, int L,
                            nef_stream_t stream);
int nef_outconv_fwd(const float* x, const float* w, const float* bias, float* out, int N, int C, int L,
                    nef_stream_t stream);
/* Variants that take the pre-BatchNorm tensor and apply x' = max(0, x*a[p][c] + b[p][c]), p = n / Bp, on the fly. */
int nef_outconv_fwd_pro(const float* x, const float* a, const float* b, int Bp, const float* w, const float* bias,
                        float* out, int N, int C, int L, nef_stream_t stream);
int nef_outconv_bwd_weight_pro(const float* gout, const float* out, const float* x, const float* a, const float* b, int Bp,
                               float* gw, float* gb, void* ws, size_t ws_bytes, int N, int C, int L, nef_stream_t stream);
int nef_outconv_bwd_data(const float* gout, const float* out, const float* w, float* gx, int N, int C, int L,
                         nef_stream_t stream);
size_t nef_outconv_bwd_weight_ws_bytes(int C);
int nef_outconv_bwd_weight(const float* gout, const float* out, const float* x, float* gw, float* gb, void* ws,
                           size_t ws_bytes, int N, int C, int L, nef_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Loss.  codes/network/loss/losses.py:21-50 (+ solver.py:185-186 noise):
 *   l1 = mean|sg(pred)-pred_p|, l2 = mean|sg(pred)-pred_l|, l3 = mean|pred-target| (or squared, reg_l2),
 *   losses[4] = {f0*l1 + f1*l2 + f2*l3, f0*l1, f1*l2, f2*l3};  use_mask bit i enables term i+1.
 * bwd: gradients wrt pred (term 3 only), pred_p, pred_l scaled by gscale (d loss). */
size_t nef_loss_ws_bytes(void);
int nef_loss_fwd(const float* pred, const float* pred_p, const float* pred_l, const float* target, float* losses,
                 void* ws, size_t ws_bytes, int64_t n, float f0, float f1, float f2, int reg_l2, int use_mask,
                 nef_stream_t stream);
int nef_loss_bwd(const float* pred, const float* pred_p, const float* pred_l, const float* target,
                 const float* gscale /* device scalar */, float* g_pred, float* g_p, float* g_l, int64_t n, float f0,
                 float f1, float f2, int reg_l2, int use_mask, nef_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * SGD with momentum over a flat buffer.  codes/solver/optim_scheduler.py:10 (torch.optim.SGD semantics:
 * first step buf = g, afterwards buf = mu*buf + g; p -= lr*buf).  g is multiplied by gscale first
 * (1/world_size after an all-reduce sum). */
int nef_sgd_momentum(float* p, const float* g, float* buf, int64_t n, float lr, float mu, float gscale,
                     int first_step, const float* skip_if_positive /* NULL, or a device word: > 0 = leave p and buf as they are
                     (a step whose gradients are tainted, see nef_h2_taint) */, int32_t* skipped /* NULL, or a device counter of
                     skipped steps */, const float* lr_dev /* NULL, or a device word that replaces `lr` at run time (hipGraph replay: a scheduler's
                     new learning rate without a re-capture) */, nef_stream_t stream);
/* The split-fp16 convolutions (conv args wino = 3, nef_conv_bwd_weight_h2) count the waves that had to clamp an operand at fp16's
 * range in a device counter (x_clamped); such a launch's results are wrong.  nef_h2_taint writes out[0] = (float)(*clamped_total -
 * *mark) -- the clamps since the previous call -- and sets *mark = *clamped_total: called once per train step behind the backward
 * pass, its output word travels with the gradients (summed by the data-parallel all-reduce, so every rank sees a clamp on any rank)
 * and makes nef_sgd_momentum skip the update.  No counterpart in the reference (its fp32 nn.Conv1d cannot overflow at 65504,
 * codes/network/model_nefnet.py:18-21); capturable, nothing is read by the host. */
int nef_h2_taint(const int32_t* clamped_total, int32_t* mark, float* out, nef_stream_t stream);
/* choice[0..1] = (c1, c2), seed[0] = seed_value, by a launch that carries the values as kernel arguments: the per-step host
 * decisions of a replayed (hipGraph) train step -- the two Standin lead draws of codes/network/model_nefnet.py:154,156 and the
 * dropout seed -- reach the device words the captured kernels read without a blocking host-to-device copy. */
int nef_step_words(int32_t* choice, int64_t* seed, int c1, int c2, int64_t seed_value, nef_stream_t stream);
/* Once per forward pass over the table of split-fp16 call-site magnitudes (ops.amax_roll; no counterpart in the reference, whose
 * fp32 convs need no operand scale): cur[i] = nxt[i] where nxt[i] > 0 and (cur[i] <= 0, or nxt[i] > follow_up * cur[i], or
 * nxt[i] * follow_down < cur[i], or follow_always); then nxt[i] = 0.  One launch. */
int nef_amax_roll(float* cur, float* nxt, int n, float follow_up, float follow_down, int follow_always, nef_stream_t stream);
/* out = concatenation of the n device tensors srcs[k] (sizes[k] floats each), one launch per 64 tensors.  `srcs` / `sizes` are HOST
 * arrays.  Builds the flat gradient buffer FusedSGD / the data-parallel all-reduce work on (replaces the torch.cat of
 * codes/solver's per-parameter .grad tensors; optim_scheduler.py:10 steps them one by one). */
int nef_flatten(const float* const* srcs, const int64_t* sizes, int n, float* out, nef_stream_t stream);
/* w [Co][2 Cih][K] -> grouped [2 Co][Cih][K] (group = input-channel half: the first decoder conv runs once per distinct half,
 * DESIGN.md section 2; weight of codes/network/model_nefnet.py:18), inverse != 0: the other way (its gradient). */
int nef_regroup_halves(const float* src, float* dst, int Co, int Cih, int K, int inverse, nef_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Polyphase form of a K = 3 conv behind the x2 linear upsampling (codes/network/model_nefnet.py:101-105, nn.Upsample + DoubleConv's
 * first conv): output 2m + p is a K = 3 conv of the HALF-resolution input with phase weights W'_p.
 * nef_poly_weights: w [rows][Cig][3] -> wsyn [2 rows][Cig][3], row 2 r + p = W'_p of row r.
 * Backward-data through conv + upsampling in one pass at half resolution: nef_conv_fwd with pro_mode 4 (input = the
 * full-resolution gradient [B][G Cog][T] read as 2 Cog phase channels of length T / 2), weights = wsyn packed transposed / flipped,
 * output = the gradient wrt the half-resolution input; then nef_poly_bwd_edge adds the two row-end terms the phase form leaves out
 * (and their share of the BatchNorm-backward sums, into slot 0 of the sample: bnb_* as in nef_conv_args, NULL slots = none). */
/* out[b][c] = sum over the nslot slots of sample b of slots[c][b*nslot + s][0] (the word-0 sums a conv epilogue left, stats_mode 1). */
int nef_slots_to_rows(const float* slots, int nslot, float* out, int B, int C, nef_stream_t stream);
int nef_poly_weights(const float* w, float* wsyn, int rows, int Cig, int tile_Cr /* 0, or the channels per group (a multiple of 64):
                     rows in the TILE order of the polyphase forward launch, phase p of channel co of group g = row
                     g 2 Cr + (co / 64) 128 + ((co / 32) & 1) 64 + p 32 + co % 32 */, nef_stream_t stream);
/* Forward in polyphase form: nef_conv_fwd with pro_mode 8 (| 1: the BatchNorm affine + ReLU prologue) -- x the half-resolution input
 * [B][G Cin_g][T], weights = tile-ordered wsyn (Cout_g = 2 x channels), y [B][G Cout_g / 2][2 T], bias / stats only; then
 * nef_poly_fwd_edge corrects the first and the last output column (and slot 0 of the statistics) for the conv's zero padding. */
int nef_poly_fwd_edge(const float* x, const float* w, float* y, int B, int G, int Cr /* output channels per group */, int Cig,
                      int T /* length of y = 2 x length of x */, const float* pro_a, const float* pro_b, int pro_Bp, float* stats,
                      int nslot, float* xedge /* NULL, or [B][G Cig][2]: the prologue's output at positions 0 and T/2 - 1 */,
                      nef_stream_t stream);
int nef_poly_bwd_edge(const float* gy, const float* w /* [G Cog][Cig][3], the conv's own weight */, float* gx, int B, int G, int Cog,
                      int Cig, int T /* length of gy = 2 x length of gx */, const float* bnb_x, const float* bnb_mean,
                      const float* bnb_invstd, const float* bnb_a, const float* bnb_b, int bnb_Bp, float* bnb_slots, int nslot,
                      int gy_phase_major /* gy stored [B][G 2 Cog][T / 2] (nef_bn_relu_bwd_phase_major) */, nef_stream_t stream);
/* PHASE-MAJOR gradients: nef_bn_relu_bwd_phase_major / nef_bn_relu_bwd_combine3_phase_major are nef_bn_relu_bwd /
 * nef_bn_relu_bwd_combine3 writing row r of their output as the two half-length rows 2 r (even positions) and 2 r + 1 (odd
 * positions) of a [.., 2 C, L / 2] tensor (L % 4 == 0 resp. L % 2 == 0).  That is the operand the polyphase backward passes of the
 * conv behind the upsampling want: backward-data = a PLAIN nef_conv_fwd over it (weights nef_poly_weights, transposed / flipped)
 * + nef_poly_bwd_edge(gy_phase_major = 1); weight gradient = nef_conv_bwd_weight_h2 with pro_mode 4 (| 1: affine prologue; bit 2 =
 * the half-resolution x window is continued with x[0] / x[T-1] at the row ends) giving gw2 [G 2 Cog][Cig][3], then
 * nef_poly_wgrad_fold: gw2 folded back onto the conv's own taps minus the row-end terms (xedge [B][G Cig][2] = the prologue's
 * output at the first / last position, written by nef_poly_fwd_edge). */
int nef_bn_relu_bwd_phase_major(const float* gy, const float* x, const float* gamma, const float* mean, const float* invstd,
                                const float* a, const float* b, float* gx, float* ggamma, float* gbeta, float* gx_chan_sum,
                                void* ws, size_t ws_bytes, int P, int Bp, int C, int L, const float* slots, int nslot,
                                nef_stream_t stream);
int nef_bn_relu_bwd_combine3_phase_major(const float* gy, const float* x, const float* mean, const float* invstd, const float* a,
                                         const float* b, float* gP2, float* ggamma, float* gbeta, float* gx_chan_sum, void* ws,
                                         size_t ws_bytes, int Bp, int C, int L, const float* slots, int nslot,
                                         nef_stream_t stream);
size_t nef_poly_wgrad_fold_ws_bytes(int B, int G, int Cog, int Cig);
int nef_poly_wgrad_fold(const float* gw2, const float* gy_pm, const float* xedge, float* gw, void* ws, size_t ws_bytes, int B, int G,
                        int Cog, int Cig, int T /* full-resolution length = 2 x the rows of gy_pm */, nef_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Test-phase metrics on the device.  Replaces PSNR / SSIM of codes/utils/mertic.py:7-32 as called from
 * codes/solver/solver.py:202-228: per (sample, view) row over the un-padded region [0, rois[i][6][0]) (whole row
 * when rois is NULL).  pred, gt fp32 [B][Q][L]; psnr, ssim fp64 [B][Q].  PSNR = 100 for an exact match, else
 * 20*log10(1/rmse).  SSIM = skimage structural_similarity(data_range=1.0) for 1-D input (7-tap uniform window,
 * sample covariance, borders of 3 cropped); NaN for rows shorter than 7 (skimage raises there). */
int nef_view_metrics(const float* pred, const float* gt, const int64_t* rois, double* psnr, double* ssim, int B, int Q,
                     int L, nef_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Half-precision panorama decoder (BASELINE configs 4/5): the eval-mode view sweep of
 * model_nefnet.py:181-190 / gen_ecg :196-218 with BatchNorm folded into the convs (nef_fold_bn), activations kept
 * as fp16 [pair][time][channel] and the four wide decoder convs (model_nefnet.py:18,21 inside :103,:105) on the
 * fp16 matrix cores with fp32 accumulation.  No reference counterpart in reduced precision: gated against the fp32
 * path (2e-3 rel-L2).  Pairs are (sample, angle), sample-major: pair n = b*nq + i.
 * nef_pano_h_from_f32 : x fp32 [B][C][T] -> y fp16 [B][T][C].
 * nef_pano_h_pack_weight: w fp32 [Cout][Cin][3] (BN already folded) -> fp16 MFMA A-fragment order, Cout*Cin*3 halfs.
 * nef_pano_h_conv     : y[n] = ReLU(conv_k3_pad1(pro(x[n / x_div])) + bias), x fp16 [.][Tin][Cin], y fp16 [N][T][Cout];
 *                       pro_mode bit0: channel ci times scale[(n/nq)*sc_bs + (n%nq)*sc_is + ci] (fp32; the query
 *                       encoding q of model_nefnet.py:184-186), bit1: x2 linear upsample along time (Tin = T/2,
 *                       nn.Upsample align_corners=False, :102,:104).  (Cin,Cout) in {(256,128),(128,128),(128,64),(64,64)}.
 * nef_pano_h_outconv  : out[(n/nq)*out_bs + (n%nq)*out_is + t] = sigmoid((conv_k3(x[n]; w [1][64][3]) + bias)/3),
 *                       x fp16 [N][T][64], out fp32 (model_nefnet.py:106 + :186). */
int nef_pano_h_from_f32(const float* x, void* y, int B, int C, int T, nef_stream_t stream);
int nef_pano_h_pack_weight(const float* w, void* wp, int Cout, int Cin, nef_stream_t stream);
int nef_pano_h_conv(const void* x, const void* wp, const float* bias, const float* scale, void* y, int N, int T, int Cin,
                    int Cout, int pro_mode, int x_div, int nq, int64_t sc_bs, int64_t sc_is, nef_stream_t stream);
 /* nef_pano_h_conv_pair: layers 1 and 2 of the decoder (model_nefnet.py:102-103: Upsample, DoubleConv(256,128)) in one
 * pass (T even; up to 256 output rows one tile per pair, longer sequences -- round 6 -- in tiles of 252 output rows with recomputed
 * halo slots):
 *   y[n] = ReLU(conv_k3(ReLU(conv_k3(scale[n] * up2(x[n / x_div])) + bias1)) + bias2),  x fp16 [.][T/2][256],
 * y fp16 [N][T][128]; the 128-channel intermediate stays on chip and is rounded to fp16 exactly as the two-call
 * sequence nef_pano_h_conv(pro_mode 3) + nef_pano_h_conv(pro_mode 0) rounds it (bit-identical results). */
/* nef_pano_h_conv_tail (round 6): layers 3 and 4, the last conv and sigmoid(x/3) of the decoder (model_nefnet.py:104-106, :186) in
 * one pass (T even; up to 512 output rows one tile per pair, longer sequences in tiles of 508 output rows with recomputed halo slots):
 *   out[(n/nq)*out_bs + (n%nq)*out_is + t] = sigmoid((conv_k3(ReLU(conv_k3(ReLU(conv_k3(up2(x[n])) + bias3)) + bias4); wout) + bout) / 3),
 * x fp16 [N][T/2][128] (layer 2's output), wp3 / wp4 = nef_pano_h_pack_weight of the folded 128->64 / 64->64 weights, wout fp32
 * [1][64][3].  The two 64-channel intermediates stay on chip, rounded to fp16 exactly as nef_pano_h_conv(pro_mode 2) +
 * nef_pano_h_conv_outconv round them. */
int nef_pano_h_conv_tail(const void* x, const void* wp3, const float* bias3, const void* wp4, const float* bias4, const float* wout,
                         const float* bout, float* out, int N, int T, int nq, int64_t out_bs, int64_t out_is, nef_stream_t stream);
int nef_pano_h_conv_pair(const void* x, const void* wp1, const float* bias1, const float* scale, const void* wp2,
                         const float* bias2, void* y, int N, int T, int x_div, int nq, int64_t sc_bs, int64_t sc_is,
                         nef_stream_t stream);
 /* nef_pano_h_conv_outconv: the 64->64 layer and the last conv in one pass: out = sigmoid((conv_k3(ReLU(conv_k3(x) +
 * bias); wout) + bout)/3); the 64-channel intermediate never reaches memory (it is rounded to fp16 exactly as the
 * two-call sequence nef_pano_h_conv + nef_pano_h_outconv rounds it). */
int nef_pano_h_conv_outconv(const void* x, const void* wp, const float* bias, const float* wout, const float* bout,
                            float* out, int N, int T, int nq, int64_t out_bs, int64_t out_is, nef_stream_t stream);
int nef_pano_h_outconv(const void* x, const float* w, const float* bias, float* out, int N, int T, int nq,
                       int64_t out_bs, int64_t out_is, nef_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NEFNET_HIP_H */
