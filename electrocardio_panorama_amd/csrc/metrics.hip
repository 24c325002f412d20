// Evaluation metrics of the test phase on the device: PSNR and SSIM per (sample, view) row on the un-padded region
// [0, rois[i][6][0]).  Replaces the host loops of reference codes/utils/mertic.py:7-32 that the test-phase body of
// Solver.run_one_epoch calls up to 2 + 2*gen_num times per batch (codes/solver/solver.py:202-228), each on a fresh
// D2H copy.  One launch fills a [B, Q] table of each metric; the solver averages table columns (generated /
// regressed / single leads) once per epoch on the host.
//
// PSNR (mertic.py:7-21): rmse over the row; 100 when rmse == 0, else 20*log10(1/rmse).
// SSIM (mertic.py:24-32 -> skimage.metrics.structural_similarity(x, y, data_range=1.0) for 1-D float input, skimage
// 0.16-0.19, not vendored by the reference): 7-tap uniform window, sample covariance (cov_norm = 7/6), K1 = 0.01,
// K2 = 0.03, the (win-1)/2 = 3 border positions cropped before averaging -- so no boundary mode is ever visible.
// Rows shorter than the window give NaN (skimage raises ValueError there).
#include "nef_common.h"

namespace {

constexpr int WIN = 7;

__global__ __launch_bounds__(256) void view_metrics_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                           const int64_t* __restrict__ rois, double* __restrict__ psnr,
                                                           double* __restrict__ ssim, int Q, int L) {
    __shared__ double sm[4];
    const int row = blockIdx.x;
    const int i = row / Q;
    int end = L;
    if (rois) {
        const int64_t e = rois[(int64_t)i * 14 + 12];       // rois[i, -1, 0]
        end = e < 0 ? 0 : (e > L ? L : (int)e);
    }
    const float* x = pred + (int64_t)row * L;
    const float* y = gt + (int64_t)row * L;
    double se = 0.0;
    for (int t = threadIdx.x; t < end; t += blockDim.x) {
        const double d = (double)x[t] - (double)y[t];
        se += d * d;
    }
    se = nef_block_sum_d(se, sm);
    double ss = 0.0;
    const double c1 = 0.01 * 0.01, c2 = 0.03 * 0.03, cov_norm = (double)WIN / (WIN - 1.0);
    for (int p = 3 + threadIdx.x; p < end - 3; p += blockDim.x) {
        double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
#pragma unroll
        for (int k = -3; k <= 3; ++k) {
            const double a = (double)x[p + k], b = (double)y[p + k];
            sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
        }
        const double ux = sx / WIN, uy = sy / WIN;
        const double vx = cov_norm * (sxx / WIN - ux * ux), vy = cov_norm * (syy / WIN - uy * uy);
        const double vxy = cov_norm * (sxy / WIN - ux * uy);
        ss += ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2));
    }
    ss = nef_block_sum_d(ss, sm);
    if (threadIdx.x == 0) {
        if (end <= 0) {
            psnr[row] = __builtin_nan("");
        } else {
            const double rmse = sqrt(se / (double)end);
            psnr[row] = rmse == 0.0 ? 100.0 : 20.0 * log10(1.0 / rmse);
        }
        ssim[row] = end >= WIN ? ss / (double)(end - (WIN - 1)) : __builtin_nan("");
    }
}

}  // namespace

extern "C" int nef_view_metrics(const float* pred, const float* gt, const int64_t* rois, double* psnr, double* ssim,
                                int B, int Q, int L, nef_stream_t stream) {
    NEF_ENTER();
    NEF_REQUIRE(pred && gt && psnr && ssim, NEF_E_NULL);
    NEF_REQUIRE(B > 0 && Q > 0 && L > 0, NEF_E_SHAPE);
    hipLaunchKernelGGL(view_metrics_kernel, dim3((unsigned)(B * Q)), dim3(256), 0, (hipStream_t)stream, pred, gt, rois,
                       psnr, ssim, Q, L);
    return nef_launch_status();
}
