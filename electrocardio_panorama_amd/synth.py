"""Seeded synthetic ECG batches with the reference's `meta` schema.

Mirrors what EcgTianChiInterval.__getitem__ hands the solver (reference
codes/dataset/tianchi.py:212-224): one heartbeat per sample, signals min-max
normalised to [0, 1] with a zero tail after the beat's end point (:109-111,
:199-211), seven contiguous int64 ROIs spanning [0, L] (:103-106), view angles
from the 12-lead table (:55-67) with Gaussian jitter (:77-82).  numpy only.
"""
import numpy as np

LEAD_THETA = np.array([
    [np.pi / 2, np.pi / 2],            # I
    [np.pi * 5 / 6, np.pi / 2],        # II
    [np.pi / 2, -np.pi / 18],          # v1
    [np.pi / 2, np.pi / 18],           # v2
    [np.pi * (19 / 36), np.pi / 12],   # v3
    [np.pi * (11 / 20), np.pi / 6],    # v4
    [np.pi * (16 / 30), np.pi / 3],    # v5
    [np.pi * (16 / 30), np.pi / 2],    # v6
    [np.pi * (5 / 6), -np.pi / 2],     # III
    [np.pi * (1 / 3), -np.pi / 2],     # aVR
    [np.pi * (1 / 3), np.pi / 2],      # aVL
    [np.pi * 1, np.pi / 2],            # aVF
])

# boundary fractions of the bundled sample beat (P on/off, R on/off, T on/off, end)
_SEG_FRAC = np.array([0.115, 0.139, 0.229, 0.307, 0.463, 0.531])

LEADS_FOR = {1: [1], 2: [1, 6], 3: [1, 3, 6], 4: [2, 6, 0, 8], 5: [2, 6, 0, 8, 10],
             8: list(range(8)), 9: [2, 4, 5, 6, 7, 8, 9, 10, 11], 12: list(range(12))}


def make_rois(rng, B, L):
    jit = 1.0 + rng.uniform(-0.1, 0.1, size=(B, 6))
    b = np.rint(L * _SEG_FRAC[None, :] * jit).astype(np.int64)
    b = np.maximum.accumulate(np.clip(b, 1, L - 1), axis=1)
    edges = np.concatenate([np.zeros((B, 1), np.int64), b, np.full((B, 1), L, np.int64)], axis=1)
    return np.stack([edges[:, :-1], edges[:, 1:]], axis=2)          # [B, 7, 2]


def _beats(rng, rois, theta, L):
    """One pseudo-beat per (sample, view): P, R, T Gaussians whose gains depend on the view angle."""
    B, n_view = theta.shape[0], theta.shape[1]
    t = np.arange(L, dtype=np.float64)[None, None, :]
    r = rois.astype(np.float64)
    centers = np.stack([(r[:, 0, 0] + r[:, 0, 1]) / 2, (r[:, 2, 0] + r[:, 2, 1]) / 2,
                        (r[:, 4, 0] + r[:, 4, 1]) / 2], axis=1)     # P, R, T
    widths = np.stack([(r[:, 0, 1] - r[:, 0, 0]) / 4 + 1, (r[:, 2, 1] - r[:, 2, 0]) / 6 + 1,
                       (r[:, 4, 1] - r[:, 4, 0]) / 4 + 1], axis=1)
    th, ph = theta[..., 0], theta[..., 1]
    gains = np.stack([0.15 * np.sin(th) * np.cos(ph / 2), 1.0 * np.cos(th - 1.0) * np.cos(ph / 3 + 0.3),
                      0.30 * np.sin(th + 0.4) * np.cos(ph / 2 - 0.2)], axis=2)   # [B, n_view, 3]
    sig = np.zeros((B, n_view, L))
    for k in range(3):
        sig += gains[:, :, k, None] * np.exp(-0.5 * ((t - centers[:, None, k, None]) / widths[:, None, k, None]) ** 2)
    sig += rng.normal(0.0, 0.01, size=sig.shape)
    return sig


def make_batch(B, V, L, seed=123, Q=0, jitter_deg=2.5, leads=None):
    """Dict of numpy arrays: data f32[B,V,L], rois i64[B,7,2], input_theta f32[B,V,2],
    target_view f32[B,L], target_theta f32[B,2], noise f32[B,L] (zeros), and when Q>0
    rest_theta f32[B,Q,2] / rest_view f32[B,Q,L]."""
    rng = np.random.default_rng(seed)
    leads = list(leads) if leads is not None else LEADS_FOR.get(V, list(range(V)))
    assert len(leads) == V
    rois = make_rois(rng, B, L)
    others = [i for i in range(12) if i not in leads] or list(range(12))
    tgt_idx = rng.choice(others, size=B)
    table = LEAD_THETA[None] + rng.normal(0.0, np.deg2rad(jitter_deg), size=(B, 12, 2))
    in_theta = table[:, leads]
    tgt_theta = table[np.arange(B), tgt_idx]
    all_theta = np.concatenate([in_theta, tgt_theta[:, None]], axis=1)
    if Q:
        k = np.arange(Q)
        rest_theta = np.stack([np.full(Q, np.pi / 2), -np.pi + 2 * np.pi * k / Q], axis=1)
        rest_theta = np.broadcast_to(rest_theta[None], (B, Q, 2))
        all_theta = np.concatenate([all_theta, rest_theta], axis=1)
    sig = _beats(rng, rois, all_theta, L)
    end = rois[:, 6, 0]
    live = np.arange(L)[None, :] < end[:, None]                      # zero tail after the beat
    lo = np.where(live[:, None, :], sig, np.inf).min(axis=(1, 2), keepdims=True)
    hi = np.where(live[:, None, :], sig, -np.inf).max(axis=(1, 2), keepdims=True)
    sig = np.where(live[:, None, :], (sig - lo) / (hi - lo), 0.0)
    out = {
        "data": sig[:, :V].astype(np.float32),
        "rois": rois,
        "input_theta": in_theta.astype(np.float32),
        "target_view": sig[:, V].astype(np.float32),
        "target_theta": tgt_theta.astype(np.float32),
        "noise": np.zeros((B, L), np.float32),
    }
    if Q:
        out["rest_theta"] = np.ascontiguousarray(rest_theta).astype(np.float32)
        out["rest_view"] = sig[:, V + 1:].astype(np.float32)
    return out
