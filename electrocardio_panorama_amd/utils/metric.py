"""Evaluation metrics of reference codes/utils/mertic.py:7-32 (host side, numpy; not on the train-step path).

PSNR follows the reference line by line in meaning: per (sample, view) on the un-padded region `[:rois[i,-1,0]]`,
`20*log10(1/rmse)` with 100 for an exact match, averaged.  SSIM in the reference is
`skimage.metrics.structural_similarity(x, y, data_range=1.0)`; scikit-image is not available in this image, so its
published algorithm (Wang et al. 2004 as implemented by scikit-image 0.16-0.19 for 1-D float input: 7-tap uniform window,
sample covariance, K1=0.01, K2=0.03, borders of (win-1)//2 cropped before averaging) is restated here -- parity for SSIM
is unpinned against skimage itself."""
import math

import numpy as np
from scipy.ndimage import uniform_filter


def PSNR(pred, gt, rois=None, shave_border=0):
    vals = []
    for i in range(pred.shape[0]):
        end = int(rois[i, -1, 0]) if rois is not None else pred.shape[2]
        for j in range(pred.shape[1]):
            d = pred[i, j, :end].astype(np.float64) - gt[i, j, :end].astype(np.float64)
            rmse = math.sqrt(float(np.mean(d ** 2)))
            vals.append(100 if rmse == 0 else 20 * np.log10(1.0 / rmse))
    return float(np.mean(vals))


def ssim_1d(x, y, data_range=1.0, win=7):
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    if x.shape[0] < win:
        raise ValueError("win_size exceeds signal extent")
    cov_norm = win / (win - 1.0)
    ux, uy = uniform_filter(x, win), uniform_filter(y, win)
    uxx, uyy, uxy = uniform_filter(x * x, win), uniform_filter(y * y, win), uniform_filter(x * y, win)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
    pad = (win - 1) // 2
    return float(s[pad:s.shape[0] - pad].mean())


def SSIM(pred, gt, rois=None):
    vals = []
    for i in range(pred.shape[0]):
        end = int(rois[i, -1, 0]) if rois is not None else pred.shape[2]
        for j in range(pred.shape[1]):
            vals.append(ssim_1d(pred[i, j, :end], gt[i, j, :end], 1.0))
    return float(np.mean(vals))
