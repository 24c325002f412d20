"""Thin tensor-level wrappers over the C ABI (include/nefnet_hip.h).

PyTorch is used only for device memory and streams: every function takes contiguous fp32 (int64 for
ROIs) tensors on a HIP device, allocates the outputs, and enqueues the library call on the current
stream.  Nothing here computes on the host and nothing falls back to torch ops.
"""
import ctypes as C
import contextlib
import itertools
import os

import torch
from . import _env

from . import _lib

N_SEG, ROI_BINS = 7, 16


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, dtype=torch.float32):
    assert t.is_cuda and t.dtype == dtype and t.is_contiguous(), (t.device, t.dtype, t.is_contiguous())
    return t


_WS = {}

# bench.py sets this to a list to collect (tag, start_event, end_event) around selected launches; the events are
# recorded on the stream the kernel is launched on, so their difference is that kernel's device time.
PROFILE = None
PROFILE_ONLY = None     # optional predicate(tag): bracket only these launches (every event pair costs ~3 us of GPU idle)
# matrix-core multiplies EXECUTED per algorithmic multiply of a tagged conv launch (tag -> fraction), filled while PROFILE
# is on: 1 direct; K=3: 2/3 through F(2,3), 1/2 through F(4,3) / the transposed F(3,4); K=7: 9/14 forward (F(2,4)+F(2,3)),
# 13/28 backward-data (F(4,4)+F(4,3)) and weight gradient (transposed F(4,4)+F(3,4)), 10/14 for the 3+3+1 forms
EXEC_FRAC = {}
# ... and fp16 matrix-core multiplies executed per algorithmic multiply (the split-fp16 kernels: 3; every other form: 0)
EXEC_FP16 = {}


def _exec_frac(K, form):
    """form: 0 direct, 1 the F(2,.) family, 2 the F(4,.) family (forward / backward-data: nef_conv_args.wino; weight
    gradient: 1 = transposed F(3,2), 2 = the round-2/3 forms ops.conv_bwd_weight calls `wino=4`)."""
    if not form or K == 1:
        return 1.0
    if K == 3:
        return 2.0 / 3.0 if form == 1 else 0.5
    return 9.0 / 14.0 if form == 1 else 13.0 / 28.0


def _timed(tag):
    if PROFILE is None or (PROFILE_ONLY is not None and not PROFILE_ONLY(tag)):
        return None
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    PROFILE.append((tag, s, e))
    s.record()
    return e


def _hbm(name, *tensors):
    """bench.py hook for the HBM-bound passes: the tag carries the launch's ALGORITHMIC bytes (its external inputs and
    outputs, each counted once), so bytes / event time is the pass's achieved bandwidth."""
    if PROFILE is None:
        return None
    return _timed(("hbm", name, sum(t.numel() * t.element_size() for t in tensors if t is not None)))


def _done(ev):
    if ev is not None:
        ev.record()


def workspace(nbytes, device):
    """Caller-owned scratch, grown on demand, reused by stream-ordered calls (one buffer per stream)."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    cur = _WS.get(key)
    if cur is None or cur.numel() < nbytes:
        cur = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = cur
    return cur


class SideStream:
    """A second HIP stream for work that is off the critical dependency chain of the backward pass (weight and bias
    gradients are only needed by the optimiser).  The MFMA-bound bwd-weight kernels then overlap with the HBM-bound
    elementwise kernels of the main chain instead of queueing behind them.

    Memory safety without `record_stream` (which makes the caching allocator hoard memory: blocks with pending
    cross-stream uses cannot be recycled, so it keeps reserving new ones): the inputs of a side-stream launch are kept
    referenced here until an event recorded behind that launch has completed, or until `join()` has made the main
    stream wait for the side stream -- in both cases any later reuse of their memory is ordered after the side
    kernel.  Outputs are allocated from the side stream's pool and are only reused by side-stream work of a later
    step, which starts with `wait_stream(main)` and therefore after their last reader."""

    _cache = {}

    def __init__(self, device):
        self.device = device
        self.stream = torch.cuda.Stream(device)
        self.pending = []          # (event, tensors kept alive)

    @classmethod
    def get(cls, device):
        key = (device.type, device.index)
        if key not in cls._cache:
            cls._cache[key] = cls(device)
        return cls._cache[key]

    def run(self, fn, *inputs):
        """Run fn() on the side stream once everything enqueued so far on the current stream is done."""
        main = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(main)
        capturing = torch.cuda.is_current_stream_capturing()
        with torch.cuda.stream(self.stream):
            out = fn()
            if not capturing:
                ev = torch.cuda.Event()
                ev.record(self.stream)
        if capturing:
            # inside a hipGraph capture the fork / join become graph edges (the side kernels are parallel branches of the
            # captured step); events cannot be polled there, so the inputs simply stay referenced until join()
            self.pending.append((None, inputs))
            return out
        self.pending.append((ev, inputs))
        while self.pending and self.pending[0][0] is not None and self.pending[0][0].query():
            self.pending.pop(0)
        return out

    def join(self):
        """Make the current stream wait for everything issued on the side stream."""
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        self.pending.clear()


class _Inline:
    """Same interface, everything on the current stream (NEF_SIDE_STREAM=0)."""

    def run(self, fn, *inputs):
        return fn()

    def join(self):
        pass


class GV:
    """Grouped view of an activation: element (b, g, c, t) at base + b*bs + g*gs + c*T + t (in floats)."""
    __slots__ = ("t", "B", "G", "Cg", "T", "bs", "gs", "off")

    def __init__(self, t, B, G, Cg, T, bs, gs, off=0):
        self.t, self.B, self.G, self.Cg, self.T, self.bs, self.gs, self.off = t, B, G, Cg, T, bs, gs, off

    @property
    def ptr(self):
        return self.t.data_ptr() + 4 * self.off

    @staticmethod
    def dense(t, G):
        _chk(t)
        B, Ct, T = t.shape
        return GV(t, B, G, Ct // G, T, Ct * T, (Ct // G) * T, 0)

    @staticmethod
    def half(t, V, which):
        """Channels [which*64, which*64+64) of every lead of a [B, 128V, T] tensor (the z1/z2 split,
        reference codes/network/model_nefnet.py:127-131)."""
        _chk(t)
        B, Ct, T = t.shape
        return GV(t, B, V, 64, T, Ct * T, 128 * T, which * 64 * T)


# ------------------------------------------------------------------ stem
def stem_fwd(x, w):
    L = _lib.load()
    _chk(x), _chk(w)
    B, V, Ln = x.shape
    y = torch.empty(B, 128 * V, Ln // 4, device=x.device, dtype=torch.float32)
    ev = _hbm("stem_fwd", x, y)
    _lib.check(L.nef_stem_fwd(_p(x), _p(w), _p(y), B, V, Ln, _stream()), "nef_stem_fwd")
    _done(ev)
    return y


def stem_bwd_weight(x, w, gy):
    L = _lib.load()
    _chk(x), _chk(w), _chk(gy)
    B, V, Ln = x.shape
    gw = torch.empty_like(w)
    n = L.nef_stem_bwd_ws_bytes(V)
    ws = workspace(n, x.device)
    ev = _hbm("stem_bwd_weight", x, gy)
    _lib.check(L.nef_stem_bwd_weight(_p(x), _p(w), _p(gy), _p(gw), _p(ws), n, B, V, Ln, _stream()), "nef_stem_bwd_weight")
    _done(ev)
    return gw


# ------------------------------------------------------------------ grouped conv
# K = 3 convs through Winograd F(2,3), K = 7 convs through F(2,4) + F(2,3) on the taps split 4 + 3 (conv_mfma.hip:
# conv_wino_kernel; "F(2,3)" below stands for this F(2,.) family) wherever a whole output tile of one
# sample exists; NEF_WINOGRAD=0 keeps every conv on the direct kernel.
_WV = _env.get("NEF_WINOGRAD", "4")
WINOGRAD = _WV != "0"
# Forward / backward-data form where the caller allows the larger tile (`f4=True`: the decoder convs): 2 = F(4,3)
# (default), 1 = F(2,3) everywhere (NEF_WINOGRAD=2).  The encoder-side convs always take F(2,3), for two measured reasons:
# (1) their inputs end in the all-zero tail of a beat; F(2,3) reproduces the reference's exact 0.0 there (every product
# feeding an output only sees that output's own receptive field), F(4,3) leaves +-1e-9 of residue whose sign then
# opens ReLU gates the reference keeps closed (705 of 512 k decisions in one block); (2) with that repaired by an
# exact-zero fix-up in the kernel (built, 255 VGPRs), the larger rounding of F(4,3) still moved the stem's weights --
# a heavily cancelling gradient -- by 3.7e-4 over the 3-step reference trajectory (bar 2e-4), and the fixed-up kernel
# was no faster than F(2,3) (1.31 vs 1.33 ms on the K=7 conv).
WINO_FWD = 1 if _WV in ("1", "2") else 2
# Winograd weight gradients (nef_conv_bwd_weight_wino4): K=3 through the transposed F(3,4), K=7 with the taps split 4 + 3 over
# two launches (transposed F(4,4) + F(3,4)).  NEF_BW_WINO4=0 / NEF_BW7_F42=0 put K=3 / K=7 back on the direct kernel.
WINO_BW4 = _env.get("NEF_BW_WINO4", "1") == "1" and _WV not in ("1", "2")
WINO_BW7 = _env.get("NEF_BW7_F42", "1") == "1" and _WV not in ("1", "2")
_WINO_PLANES = {(1, 3): 4, (1, 7): 10, (2, 3): 6, (2, 7): 13}


# Split-fp16 direct convolution (csrc/conv_h2.hip, conv args wino = 3): both fp32 operands split into two fp16 terms each (22-23 bits kept),
# three fp16 matrix instructions per 16 channels and tap, fp32 accumulation -- fp32-class results at 3/16 of the fp32 matrix
# instructions' pipe time.  NEF_H2=0 keeps the fp32 Winograd forms; NEF_H2=1 takes it wherever the shape allows
# (128-channel output tiles, 16-channel input chunks, T even and >= 128; K = 7 without an input prologue).
H2 = _env.get("NEF_H2", "1") == "1"
_H2_DIR = {False: _env.get("NEF_H2_FWD", "1") == "1", True: _env.get("NEF_H2_BWD", "1") == "1"}     # diagnostics
_H2_K = _env.get("NEF_H2_K", "1,3,7").split(",")
_H2_64 = _env.get("NEF_H2_64", "1") == "1"
_H2_MIN_T = int(_env.get("NEF_H2_MIN_T", "0"))      # shortest sequence the split-fp16 kernels take (256-column tiles)
_H2_W = _env.get("NEF_H2_W", "1") == "1"          # weight gradients on the split-fp16 kernel too (csrc/conv_h2w.hip)
_H2_WK = _env.get("NEF_H2_WK", "1,3,7").split(",")
_H2_AMAX = _env.get("NEF_H2_AMAX", "sticky")      # diagnostics: "anon" = every launch measures first, "follow" = no stickiness


# Small problems stay on the fp32 kernels: the split-fp16 kernels tile a sample in 256 outputs x 128 (64) channels, and below
# ~one workgroup per CU the half-empty tiles and the per-launch setup cost more than the matrix time they save (reference-native
# batch 32 x L 512: 3.11 ms per captured step with them, 2.72 without).  The engine announces the batch of the pass it is about
# to run (BATCH_HINT); without a hint (bare ops calls) the shape rules alone decide.
BATCH_HINT = None
_H2_PACK = _env.get("NEF_H2_PACK", "1") == "1"
_H2_MIN_WGS = int(_env.get("NEF_H2_MIN_WGS", "256"))


def _h2_packed(K, T_out, pro=0):
    """Short rows (8 <= T <= 64, T % 4 == 0) of plain K = 1 / K = 3 launches: several samples per tile of conv_h2_kernel
    (csrc/conv_h2.hip, PACK).  NEF_H2_PACK=0: they stay on the fp32 kernels."""
    return _H2_PACK and K in (1, 3) and not pro and 8 <= T_out <= 64 and T_out % 4 == 0


def _h2_fills(G, Cout_g, T_out, tile_t, tile_c):
    if BATCH_HINT is None:
        return True
    if T_out < 128:       # packed short rows: (256 + 4) // (T + 4) samples per tile, 64-channel tiles
        spt = (256 + 4) // (T_out + 4)
        return G * ((Cout_g + 63) // 64) * ((BATCH_HINT + spt - 1) // spt) >= _H2_MIN_WGS
    return G * ((Cout_g + tile_c - 1) // tile_c) * BATCH_HINT * ((T_out + tile_t - 1) // tile_t) >= _H2_MIN_WGS


def h2_ok(K, Cin_g, Cout_g, T_out, pro=0):
    # NEF_H2_64=0 leaves the 64-channel output tiles (conv_h2_kernel<., ., 1>) to the F(4,3) kernels
    return (H2 and (K == 3 or (K in (1, 7) and not pro)) and T_out % 2 == 0 and
            (T_out >= max(128, _H2_MIN_T) or (_h2_packed(K, T_out, pro) and _H2_64)) and Cin_g % 16 == 0 and
            Cout_g % (64 if _H2_64 else 128) == 0)


# Input magnitudes of the split-fp16 launches, per call site (= per weight tensor and direction): `cur` is what a launch derives
# its power-of-two input scale from, `nxt` is what it max-accumulates its own operand's magnitude into; amax_roll() -- once per
# forward pass -- moves nxt into cur.  A site's FIRST launch runs twice: once to measure, once with the measured scale.  Nothing is
# read back by the host, so the launches stay capturable.
# Range, in ONE place (bench.py and Solver quote these): the scale puts `cur` at [2^8, 2^9); amax_roll follows a
# measurement UP as soon as it exceeds H2_FOLLOW_UP x cur and DOWN only once it is H2_FOLLOW_DOWN x smaller (sticky: repeated
# passes split their operands identically), so the operand a launch meets is below 2^10 as long as it grew less than
# H2_HEADROOM = 2^16 / 2^10 = 64 x since the previous pass.  Beyond that the kernels RESCUE the launch: a workgroup whose tile (weight
# gradient: share) does not fit redoes it with the scale its own data asks for (conv_h2.hip / conv_h2w.hip), so finite operands are
# never clamped; a launch that meets non-finite data (or the opt-in producer / consumer form, which still clamps) counts itself in
# `clamped`, and the train step that contains it is SKIPPED on the device (h2_taint / sgd_momentum).
H2_FOLLOW_UP = 2.0
H2_FOLLOW_DOWN = 64.0
H2_HEADROOM = 64
# Heavy-tail guard (round 6): the format keeps 22-23 bits of an element only down to H2_TAIL_WINDOW = 2^-11 of its tensor's largest
# (one power-of-two scale per tensor).  A call site counts itself in `tail` when more than H2_TAIL_FRAC = 90 % of its operand's
# nonzero elements lie below that line: practically the whole tensor sits outside the full-precision window, and the products of two
# such operands (a weight gradient) lose per-element precision (test_conv_h2_operand_distributions[lognormal]: exp(4 N(0,1)) has
# 99.4 % there; profiles/r06_h2_scale_granularity.md).  There is no sharp line -- the format is fp32-class on the NORM whatever the
# distribution -- and this model's own tensors are not far from it: the worst site of configs[1] (a gradient: a few large entries, a
# long tail) has 74 % of its nonzero elements below the window, carrying 0.26 % of its energy (`h2_tail_worst` in the bench line;
# full-size gradient parity 3.3e-5 on the flat norm, every tensor <= 1e-3).  An amax / rms ratio (the review's suggestion) fires on
# neither: the rms of a heavy-tailed sample is dominated by its largest elements (log-normal: 100-500, not > 2048).  Checked ONCE per
# site, where it measures its operand (its first launch; never inside a captured step); bench.py and Solver report the count
# (h2_tail_sites) -- a model that counts here wants NEF_H2=0.
H2_TAIL_WINDOW = 2.0 ** -11
H2_TAIL_FRAC = 0.9
AMAX_SITES = 16384
_AMAX = {}
# A call site = (scope, weight address, direction, role, batch, length).  The scope is the owning model's token
# (Model_nefnet sets it around every engine call; a fresh model or a loaded checkpoint gets a fresh token, so nothing is inherited
# from whatever lived at the same addresses before).  Without a scope (bare engine / ops calls) every launch measures first.
AMAX_SCOPE = None
_SCOPE_IDS = itertools.count(1)


def new_amax_scope():
    return next(_SCOPE_IDS)


@contextlib.contextmanager
def amax_scope(token):
    global AMAX_SCOPE
    prev, AMAX_SCOPE = AMAX_SCOPE, token
    try:
        yield
    finally:
        AMAX_SCOPE = prev


def _amax_state(dev):
    st = _AMAX.get(dev)
    if st is None:
        st = _AMAX[dev] = dict(cur=torch.zeros(AMAX_SITES, device=dev, dtype=torch.float32),
                               nxt=torch.zeros(AMAX_SITES, device=dev, dtype=torch.float32), index={}, ready=set(), used=False, occ={}, n=0,
                               clamped=torch.zeros(1, device=dev, dtype=torch.int32),       # waves that clamped (device total, never reset)
                               mark=torch.zeros(1, device=dev, dtype=torch.int32),          # ... at the last step boundary (h2_taint)
                               skipped=torch.zeros(1, device=dev, dtype=torch.int32),       # train steps skipped because of a clamp
                               tail=torch.zeros(1, device=dev, dtype=torch.int32),          # call sites whose operand is heavy-tailed (H2_TAIL_*)
                               tail_stat=torch.zeros(2, device=dev, dtype=torch.float32),   # diagnostics: largest (count, energy) fraction below the window seen at a site
                               seen=0, seen_skipped=0, seen_tail=0)                                       # ... as of the host's last h2_clamped() / h2_skipped()
    return st


def h2_clamped(reset=True):
    """Waves of split-fp16 launches (sited ones: measuring launches are not counted) that had to clamp an operand element at
    fp16's range since the last call: since round 5 that means NON-FINITE data (finite operands beyond the headroom are rescued
    inside the launch), or the opt-in producer / consumer form, which has no rescue.  Those launches' results are off; a train step
    that contains one is skipped (h2_taint).  Reading synchronises: call it at a logging interval, not per step."""
    n = 0
    for st in _AMAX.values():
        tot = int(st["clamped"].item())
        n += tot - st["seen"]
        if reset:
            st["seen"] = tot
    return n


def h2_skipped(reset=True):
    """Train steps whose parameter update was skipped on the device because a split-fp16 launch of the step clamped (on this or,
    data parallel, on any rank) since the last call.  Synchronises like h2_clamped()."""
    n = 0
    for st in _AMAX.values():
        tot = int(st["skipped"].item())
        n += tot - st["seen_skipped"]
        if reset:
            st["seen_skipped"] = tot
    return n


def h2_tail_sites(reset=True):
    """Split-fp16 call sites whose operand, when the site measured it, had more than H2_TAIL_FRAC of its nonzero elements below
    H2_TAIL_WINDOW x its largest (since the last call).  Synchronises like h2_clamped()."""
    n = 0
    for st in _AMAX.values():
        tot = int(st["tail"].item())
        n += tot - st["seen_tail"]
        if reset:
            st["seen_tail"] = tot
    return n


def _note_tail(st, i, n, *tensors):
    """At a site's measuring launch (eager, once per site): slots i .. i + n - 1 of `nxt` hold the operands' amax; count the site if
    an operand has more than H2_TAIL_FRAC of its nonzero elements below H2_TAIL_WINDOW x that amax (the energy fraction there is kept
    for the diagnostics, h2_tail_stats).  A handful of torch passes per SITE LIFETIME (the tensor as stored: prologues and channel scales are not applied), nothing per
    step, nothing read by the host."""
    hit = None
    for k, t in enumerate(tensors[:n]):
        if t is None or t.numel() == 0:
            continue
        a = t.detach().abs()
        small = (a > 0) & (a < st["nxt"][i + k] * H2_TAIL_WINDOW)
        cf = small.sum().to(torch.float32) / (a > 0).sum().clamp_min(1).to(torch.float32)
        a2 = a.double() * a.double()
        ef = ((a2 * small).sum() / a2.sum().clamp_min(1e-300)).to(torch.float32)
        h = cf > H2_TAIL_FRAC
        hit = h if hit is None else (hit | h)
        st["tail_stat"][0] = torch.maximum(st["tail_stat"][0], cf)
        st["tail_stat"][1] = torch.maximum(st["tail_stat"][1], ef)
    if hit is not None:
        st["tail"] += hit.to(torch.int32)


def h2_tail_stats():
    """Diagnostics: the largest (count fraction, energy fraction) below the window any site's operand had when it was measured."""
    out = [0.0, 0.0]
    for st in _AMAX.values():
        v = st["tail_stat"].tolist()
        out = [max(out[0], v[0]), max(out[1], v[1])]
    return out


def h2_taint(out):
    """out[0] (a one-element fp32 device view, e.g. the word in front of the flat gradient buffer) = the number of waves that clamped
    since the previous call (or h2_rebase); stream-ordered, capturable.  The optimiser passes the word to sgd_momentum(skip=...).
    ONE call per optimiser step: the call advances the mark, so a second parameter group must reuse the first one's word."""
    st = _amax_state(out.device)
    _lib.check(_lib.load().nef_h2_taint(_p(st["clamped"]), _p(st["mark"]), _p(out), _stream()), "nef_h2_taint")


def h2_rebase(device=None):
    """The taint mark := the clamp counter (stream-ordered): clamps counted so far -- a test-phase forward, gen_ecg between train
    steps -- are not charged to the next train step.  Solver calls it where it has just read (and judged) the counters."""
    for dev, st in _AMAX.items():
        if device is None or torch.device(device) == dev:
            st["mark"].copy_(st["clamped"])


def _amax_index(st, site, n):
    """First of the `n` consecutive magnitude slots of a call site (allocated on first use)."""
    i = st["index"].get(site)
    if i is None:
        if st["n"] + n > AMAX_SITES:       # models come and go (tests): start over
            assert not torch.cuda.is_current_stream_capturing()
            st["index"].clear(), st["ready"].clear(), st["cur"].zero_(), st["nxt"].zero_()
            st["n"] = 0
            st["gen"] = st.get("gen", 0) + 1       # captured graphs hold the old slots: GraphedTrainStep re-captures (amax_generation)
        i = st["index"][site] = st["n"]
        st["n"] += n
    return i


def h2_export(scope_id, ptr_names):
    """The operand-magnitude slots of a model's call sites as a checkpointable blob (CPU tensors and plain tuples): the keys of
    _amax_index with the owning model's scope token dropped and the weight ADDRESSES replaced by parameter names (`ptr_names`:
    {data_ptr: name}).  With it a restored run splits its operands with the scales the uninterrupted run would have used, i.e.
    continues bit for bit (h2_import); without it a restored model measures again (equal arithmetic, a different power-of-two operand
    scale wherever a magnitude sits near a binade edge).  Sites through tensors that are not parameters (BatchNorm-folded
    panorama weights) are left out: they measure again.  Reads the device (synchronises)."""
    keys, cur, nxt = [], [], []
    for st in _AMAX.values():
        c_, n_ = st["cur"].cpu(), st["nxt"].cpu()
        for key, i in st["index"].items():
            if i not in st["ready"] or len(key) != 6 or not isinstance(key[0], tuple) or key[0][0] != scope_id:
                continue
            name = ptr_names.get(key[1][0])
            if name is None:
                continue
            n = 2 if key[2] == "conv_bwd_weight" else 1
            keys.append((bool(key[0][1]), (name, key[1][1]), key[2], int(key[3]), int(key[4]), int(key[5])))
            cur.append([float(c_[i + j]) for j in range(n)])
            nxt.append([float(n_[i + j]) for j in range(n)])
    return {"version": 1, "keys": keys, "cur": cur, "nxt": nxt}


def h2_import(scope_id, name_ptrs, blob, device):
    """Hands the slots of h2_export back to the call sites of the model that now owns `scope_id` (`name_ptrs`: {name: data_ptr});
    entries whose parameter is gone are skipped (they measure)."""
    if not blob or blob.get("version") != 1:
        return 0
    st = _amax_state(torch.device(device))
    done = 0
    for key, c_, n_ in zip(blob["keys"], blob["cur"], blob["nxt"]):
        ptr = name_ptrs.get(key[1][0])
        if ptr is None:
            continue
        full = ((scope_id, bool(key[0])), (ptr, key[1][1])) + tuple(key[2:])
        i = _amax_index(st, full, len(c_))
        st["cur"][i:i + len(c_)] = torch.tensor(c_, dtype=torch.float32)
        st["nxt"][i:i + len(n_)] = torch.tensor(n_, dtype=torch.float32)
        st["ready"].add(i)
        done += 1
    if done:
        st["used"] = True
    return done


def amax_roll():
    """Once per pass (stream-ordered, a handful of tiny launches): a site's reference magnitude `cur` follows what its last launch
    measured (`nxt`) UP as soon as that exceeds H2_FOLLOW_UP x cur and DOWN once it fell below cur / H2_FOLLOW_DOWN -- the scale is
    STICKY inside that window, so passes over data of similar magnitude (a repeated step, train after eval, eager and captured
    steps of one run) split their operands identically and stay bit-reproducible, and never lags a growing operand by more than
    H2_FOLLOW_UP: H2_HEADROOM x of growth per pass is absorbed whatever the history (round 4 followed up only at 64 x, which left
    2 x in the worst case).  Elements within 2^-9 / H2_FOLLOW_DOWN of the largest keep full precision."""
    for st in _AMAX.values():
        st["occ"].clear()
        if st["used"]:
            # one launch over the slots handed out so far (rounds 4-5: ~10 torch elementwise launches per pass)
            with torch.cuda.device(st["cur"].device):
                _lib.check(_lib.load().nef_amax_roll(_p(st["cur"]), _p(st["nxt"]), int(st["n"]), H2_FOLLOW_UP, H2_FOLLOW_DOWN,
                                                     int(_H2_AMAX == "follow"), _stream()), "nef_amax_roll")
            st["used"] = False


def amax_snapshot(dev):
    """The magnitudes of the call sites handed out so far (GraphedTrainStep: around its eager probe)."""
    st = _amax_state(torch.device(dev))
    n = int(st["n"])
    return n, st["cur"][:n].clone(), st["nxt"][:n].clone(), st.get("gen", 0)


def amax_restore(dev, snap):
    """Puts the sites that existed at amax_snapshot() back to what they held then (sites created since keep what they measured)."""
    n, cur, nxt, gen = snap
    st = _amax_state(torch.device(dev))
    if n == 0 or st.get("gen", 0) != gen:
        return
    st["cur"][:n].copy_(cur)
    st["nxt"][:n].copy_(nxt)
    st["used"] = True


def amax_generation(dev):
    """Bumped whenever the site table of `dev` started over: pointers a captured launch baked in may now belong to other sites."""
    return _amax_state(dev).get("gen", 0)


def amax_move(old_ptr, new_ptr):
    """A weight tensor moved (an optimiser re-pointed its parameters into a flat buffer): its call sites keep their history."""
    for st in _AMAX.values():
        for site in [k for k in st["index"] if k[0] is not None and k[1][0] == old_ptr]:
            st["index"][(site[0], (new_ptr, site[1][1])) + site[2:]] = st["index"].pop(site)


def _packed_floats(wino, K, G, Cog, Cig, flip):
    """fp32 words of a packed operand: plain K per (co, ci); Winograd: planes per (co, ci); split-fp16: two halves per weight
    = K words, + one descale word per output row of the launch."""
    if wino == 3:
        return G * Cog * Cig * K + G * (Cig if flip else Cog)
    return G * Cog * Cig * (_WINO_PLANES[(wino, K)] if wino else K)


def wino_ok(K, Cin_g, Cout_g, T_out, pro=0):
    return (WINOGRAD and (K == 3 or (K == 7 and not pro)) and T_out % 2 == 0 and Cin_g % 16 == 0 and
            ((Cout_g % 128 == 0 and T_out >= 128) or (Cout_g % 128 != 0 and Cout_g % 64 == 0 and T_out >= 256)))


_PREPACKED = {}     # (weight data_ptr, G, flip, wino, src) -> operand packed by pack_many(), consumed by the next pack_weight()


def _logical_shape(w, src):
    """Shape of the weight tensor that is packed: w's own, or -- src = ("poly", Cr) -- the PHASE weights of conv1d(upsample2(x), w)
    (twice the rows, formed inside the pack kernel: nef_pack_desc.src_mode 1)."""
    if src is None:
        return tuple(w.shape)
    assert src[0] == "poly" and w.shape[2] == 3, src
    return (2 * w.shape[0], w.shape[1], 3)


def _pack_shape(w, G, flip, T, f4=False, plain=True, src=None):
    """`plain`: the launch has no prologue, channel scale, statistics or BatchNorm-backward sums -- the only launches the packed
    short-row form of the split-fp16 kernel (8 <= T <= 64) takes; pass False for the others so that they fall back to the fp32 kernels."""
    shp = _logical_shape(w, src)
    Cog, Cig, K = shp[0] // G, shp[1], shp[2]
    cin_g, cout_g = (Cog, Cig) if flip else (Cig, Cog)          # roles in the launch that consumes the operand
    wino = (WINO_FWD if f4 else 1) if (T is not None and wino_ok(K, cin_g, cout_g, T)) else 0
    if (T is not None and (T >= 128 or plain) and h2_ok(K, cin_g, cout_g, T) and _H2_DIR[bool(flip)] and str(K) in _H2_K and
            _h2_fills(G, cout_g, T, 256, 128 if cout_g % 128 == 0 else 64)):
        wino = 3
    return Cog, Cig, K, wino


def _req(r):
    """(w, G, flip, T[, f4[, src[, plain[, site]]]]) -> the full tuple."""
    w, G, flip, T, *rest = r
    rest = list(rest) + [False, None, True, None][len(rest):]
    return (w, G, bool(flip), T, bool(rest[0]), rest[1], bool(rest[2]), rest[3])


def pack_many(requests):
    """All operands of a pass -- or, from engine.forward(save=True), of a whole train step: forward AND backward-data operands,
    the polyphase ones included -- in ONE launch.  `requests`: iterable of (w, G, flip, T[, f4[, src[, plain[, site]]]]) exactly as
    the later pack_weight(w, G, flip=flip, T=T, f4=f4, src=src, plain=plain, site=site) calls will ask for them; those calls then
    return the pre-packed operand instead of launching.  Anything not pre-packed still packs on demand, so a missing or surplus
    request costs time, never correctness.  A call whose requests are ALL still waiting in the table (a pass-level call inside a
    step whose step-level call has already packed everything) does nothing; any other call resets the table first."""
    L = _lib.load()
    reqs, keys = [], []
    for r in requests:
        w, G, flip, T, f4, src, plain, site = _req(r)
        _chk(w)
        Cog, Cig, K, wino = _pack_shape(w, G, flip, T, f4, plain, src)
        key = (w.data_ptr(), G, flip, wino, src)
        if key not in keys:
            keys.append(key)
            reqs.append((key, w, G, Cog, Cig, K, flip, wino, src, site))
    if keys and all(_PREPACKED.get(k) is not None and _PREPACKED[k][1] is r_[1] and _PREPACKED[k][2] == r_[1]._version
                    for k, r_ in zip(keys, reqs)):
        return
    _PREPACKED.clear()
    if not reqs:
        return
    sizes = [(_packed_floats(wino, K, G, Cog, Cig, flip) + 3) // 4 * 4 for _, _, G, Cog, Cig, K, flip, wino, _, _ in reqs]      # 16-byte aligned operands
    arena = torch.empty(sum(sizes), device=reqs[0][1].device, dtype=torch.float32)
    descs = (_lib.PackDesc * len(reqs))()
    off = 0
    for d, n, (key, w, G, Cog, Cig, K, flip, wino, src, site) in zip(descs, sizes, reqs):
        if src is not None and wino != 3:
            raise _lib.NefLibraryError("pack_many: a synthesized (polyphase) operand outside the split-fp16 kernel")
        wp = arena[off:off + n]
        off += n
        if wino:
            wp.nef_wino = wino
            wp.nef_site = (w.data_ptr() if site is None else site, flip)
        d.w, d.wp, d.G, d.Cog, d.Cig, d.K, d.transpose_flip, d.wino = w.data_ptr(), wp.data_ptr(), G, Cog, Cig, K, int(flip), int(wino)
        d.src_mode, d.src_Cr = (1, int(src[1])) if src is not None else (0, 0)
        _PREPACKED[key] = (wp, w, w._version)          # keep the source alive while its pointer is the key
    _lib.check(L.nef_pack_weights(descs, len(reqs), _stream()), "nef_pack_weights")


def pack_weight(w, G, flip=False, T=None, f4=False, site=None, shared=False, plain=True, src=None):
    """w [G*Cog, Cig, K] -> packed operand (forward: [G][K][Cig][Cog]; flip: [G][K][Cog][Cig], taps reversed).
    `T`: output length of the conv launch this operand is for; when the Winograd F(2,3) path applies to that launch
    the operand is packed for it (marked with `.nef_wino`) and `conv()` takes that path; `f4` allows the F(4,3) form.
    `src` = ("poly", Cr): pack the phase weights of conv1d(upsample2(x), w) instead of w itself (split-fp16 operands only)."""
    L = _lib.load()
    _chk(w)
    Cog, Cig, K, wino = _pack_shape(w, G, flip, T, f4, plain, src)
    hit = _PREPACKED.pop((w.data_ptr(), G, bool(flip), wino, src), None)
    if hit is not None and hit[1] is w and hit[2] == w._version:
        return hit[0]
    # `site`: identity of the call site for the split-fp16 input-magnitude slots when `w` is a temporary (default: w's address)
    if wino == 3:
        wp = torch.empty(_packed_floats(3, K, G, Cog, Cig, flip), device=w.device, dtype=torch.float32)
        if src is None:
            _lib.check(L.nef_pack_weight_h2(_p(w), _p(wp), G, Cog, Cig, K, int(flip), _stream()), "nef_pack_weight_h2")
        else:
            d = (_lib.PackDesc * 1)()
            d[0].w, d[0].wp, d[0].G, d[0].Cog, d[0].Cig, d[0].K, d[0].transpose_flip, d[0].wino = w.data_ptr(), wp.data_ptr(), G, Cog, Cig, K, int(flip), 3
            d[0].src_mode, d[0].src_Cr = 1, int(src[1])
            _lib.check(L.nef_pack_weights(d, 1, _stream()), "nef_pack_weights")
        wp.nef_wino = 3
        wp.nef_site = (w.data_ptr() if site is None else site, bool(flip))
        # shared: every launch of a pass through this operand is ONE call site (a loop over chunks of one tensor family, e.g.
        # the panorama sweep's angle chunks) instead of one site per occurrence
        wp.nef_shared = bool(shared)
        return wp
    if src is not None:
        raise _lib.NefLibraryError("pack_weight: a synthesized (polyphase) operand outside the split-fp16 kernel")
    if wino:
        wp = torch.empty(G * _WINO_PLANES[(wino, K)] * Cog * Cig, device=w.device, dtype=torch.float32)
        fn = L.nef_pack_weight_wino if wino == 1 else L.nef_pack_weight_wino4
        _lib.check(fn(_p(w), _p(wp), G, Cog, Cig, K, int(flip), _stream()), "nef_pack_weight_wino")
        wp.nef_wino = wino
        return wp
    wp = torch.empty(w.numel(), device=w.device, dtype=torch.float32)
    _lib.check(L.nef_pack_weight(_p(w), _p(wp), G, Cog, Cig, K, int(flip), _stream()), "nef_pack_weight")
    return wp


def conv_stats_buffer(wp, B, G, Cog, T_out, device):
    """(slots tensor, slots per sample) for conv(..., stats=...) -- or None when the conv would not run on the F(4,3)
    kernel, whose epilogue is the one that leaves the BatchNorm slot sums."""
    wino = int(getattr(wp, "nef_wino", 0))
    if wino not in (2, 3) or (wino == 3 and T_out < 128):      # (packed short rows leave no statistics)
        return None
    # a slot = one wave's 128 columns; the split-fp16 kernel tiles a sample in 256-column workgroups (2 slots each)
    nslot = 2 * ((T_out + 255) // 256) if wino == 3 else _lib.load().nef_conv_stats_slots(T_out, Cog)
    if nslot <= 0:
        return None
    return torch.empty(G * Cog, B * nslot, 2, device=device, dtype=torch.float32), nslot


def bn_stats_from_slots(stats, gamma, beta, running_mean, running_var, P, N, Ln, eps=1e-5, momentum=0.1, nbt=None):
    """bn_train_stats from the slot sums a conv epilogue left (x [N, C, Ln] itself is not read).  `nbt`: the BatchNorm's
    num_batches_tracked (int64[]), incremented by P in the same launch."""
    L = _lib.load()
    slots, nslot = stats
    Ct = slots.shape[0]
    mean, invstd, a, b = (torch.empty(P, Ct, device=slots.device, dtype=torch.float32) for _ in range(4))
    n = L.nef_bn_ws_bytes(P, Ct)
    ws = workspace(n, slots.device)
    _lib.check(L.nef_bn_stats_from_slots(_p(slots), nslot, _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(mean),
                                         _p(invstd), _p(a), _p(b), _p(ws), n, P, N // P, Ct, Ln, eps, momentum, _p(nbt), _stream()),
               "nef_bn_stats_from_slots")
    return mean, invstd, a, b


def conv(xv, wp, Cog, K, out=None, bias=None, in_scale=None, res=None, gate=None, gate_scale=1.0, relu=False,
         mask=None, drop_p=0.0, drop_scale=1.0, seed=0, role="conv_fwd", pro=None, seed_dev=None, stats=None, bnb=None,
         x_scale=0.0, tag_extra="", res_scale=None, gate_rowscale=None, stats_mode=0):
    """out = epilogue(conv1d(prologue(x) * in_scale, w) + bias + res * res_scale).  `xv`, `res`, `gate`, `out` are GV views;
    `in_scale` / `res_scale` are (tensor, batch_stride, group_stride).  `pro` = (mode, a, b, Bp): input prologue applied while
    staging -- bit0 BatchNorm affine + ReLU with a/b [P, C_in], bit1 x2 linear upsampling of a half-resolution input
    (the output is then 2*xv.T long).  `stats`: a conv_stats_buffer() the epilogue fills with the per-slot sum and sum
    of squares of the outputs; `bnb` = (x, mean, invstd, a, b, Bp, conv_stats_buffer()[, up]): the epilogue leaves the
    BatchNorm-backward sums sum(g*m), sum(g*m*xhat) of the layer this backward-data launch propagates into.  Returns the
    output tensor."""
    L = _lib.load()
    T_out = xv.T * 2 if (pro is not None and pro[0] & 2) else xv.T
    if out is None:
        y = torch.empty(xv.B, xv.G * Cog, T_out, device=xv.t.device, dtype=torch.float32)
        out = GV.dense(y, xv.G)
    a = _lib.ConvArgs()
    a.x, a.wp, a.y = xv.ptr, _p(wp), out.ptr
    a.bias = _p(bias)
    a.in_scale = None
    if in_scale is not None:
        a.in_scale, a.sc_bs, a.sc_gs = _p(in_scale[0]), in_scale[1], in_scale[2]
    a.res = res.ptr if res is not None else None
    a.gate = gate.ptr if gate is not None else None
    a.mask = _p(mask)
    a.x_bs, a.x_gs, a.y_bs, a.y_gs = xv.bs, xv.gs, out.bs, out.gs
    if res is not None:
        a.res_bs, a.res_gs = res.bs, res.gs
    if gate_rowscale is not None:      # y = gate > 0 ? y * gate_scale * gate_rowscale[b, g, c] : 0 (split-fp16 launches only)
        assert gate is not None
        a.gate_rowscale, a.gr_bs, a.gr_gs = _p(gate_rowscale[0]), gate_rowscale[1], gate_rowscale[2]
    a.stats_mode = int(stats_mode)      # 1: `stats` receives sum_t (ungated output) x gate per slot (chscale_bwd's per-channel sums)
    if res_scale is not None:      # (tensor, batch stride, group stride) like in_scale: y += res * res_scale[b, g, c]; split-fp16 launches only
        assert res is not None
        a.res_scale, a.rs_bs, a.rs_gs = _p(res_scale[0]), res_scale[1], res_scale[2]
    if gate is not None:
        a.gate_bs, a.gate_gs = gate.bs, gate.gs
    a.B, a.T, a.G, a.Cin_g, a.Cout_g, a.K = xv.B, T_out, xv.G, xv.Cg, Cog, K
    if pro is not None and pro[0]:
        a.pro_mode, a.pro_a, a.pro_b, a.pro_Bp = pro[0], _p(pro[1]), _p(pro[2]), pro[3]
    a.relu = int(relu)
    a.gate_scale, a.drop_scale, a.drop_p, a.rng_seed = gate_scale, drop_scale, drop_p, seed
    a.rng_seed_dev = _p(seed_dev)
    a.wino = int(getattr(wp, "nef_wino", 0))
    a.x_scale = float(x_scale)
    a.stats = _p(stats[0]) if stats is not None else None
    if bnb is not None:      # (x, mean, invstd, a, b, Bp, slots buffer): BatchNorm-backward sums of the layer below
        a.bnb_x, a.bnb_mean, a.bnb_invstd, a.bnb_a, a.bnb_b = (_p(t) for t in bnb[:5])
        a.bnb_Bp, a.bnb_slots = bnb[5], _p(bnb[6][0])
        a.bnb_up = int(bnb[7]) if len(bnb) > 7 else 0
    if a.wino == 3 and not x_scale:
        st = _amax_state(xv.t.device)
        ws = getattr(wp, "nef_site", None)
        site = None
        if AMAX_SCOPE is not None and ws is not None and _H2_AMAX != "anon":
            # + the occurrence within the pass: the k-th launch through one weight keeps its own history, so a step repeated on
            # the same data finds the scales it left (bit-identical results)
            occ = st["occ"]
            k = occ[(AMAX_SCOPE, ws, role, xv.B, T_out)] = occ.get((AMAX_SCOPE, ws, role, xv.B, T_out), 0) + 1
            site = (AMAX_SCOPE, ws, role, xv.B, T_out, 0 if getattr(wp, "nef_shared", False) else k)
        i = _amax_index(st, site if site is not None else (None, "conv"), 1)      # (unscoped launches: a slot of their own, never a site's)
        a.x_amax_next = st["nxt"].data_ptr() + 4 * i
        if i not in st["ready"] or site is None:      # first launch of the site (or no scope): measure, then run
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("split-fp16 conv: a call site's first launch (it measures its operand) cannot be captured; "
                                   "run the step once eagerly first (GraphedTrainStep does)")
            st["nxt"][i] = 0.0
            _lib.check(L.nef_conv_fwd(C.byref(a), _stream()), "nef_conv_fwd")
            st["cur"][i] = st["nxt"][i]
            if site is not None:
                _note_tail(st, i, 1, xv.t)
            st["ready"].add(i)
        a.x_amax = st["cur"].data_ptr() + 4 * i
        a.x_clamped = st["clamped"].data_ptr()
        st["used"] = True
    # bench.py prices a launch by its tag; the optional 8th element says which operands are NOT at the output's resolution / are extra:
    # "up" = the input is the half-resolution tensor (x2-upsampling prologue), "bnb" / "bnbup" = the epilogue also reads the
    # BatchNorm input of the layer below (full / half resolution) for the backward sums
    extra = ("up" if (pro is not None and pro[0] & 2) else "") + ("ph" if (pro is not None and pro[0] == 4) else "") + ("pf" if (pro is not None and pro[0] & 8) else "") + tag_extra + ("bnb" + ("up" if len(bnb) > 7 and bnb[7] else "") if bnb is not None else "")
    tag = (role, K, xv.G, xv.Cg, Cog, xv.B, T_out) + ((extra,) if extra else ())
    ev = _timed(tag)
    if ev is not None:
        EXEC_FRAC[tag] = 0.0 if a.wino == 3 else _exec_frac(K, a.wino)
        EXEC_FP16[tag] = 3.0 if a.wino == 3 else 0.0
    _lib.check(L.nef_conv_fwd(C.byref(a), _stream()), "nef_conv_fwd")
    if ev is not None:
        ev.record()
    return out.t


# ------------------------------------------------------------------ polyphase form of conv1d(upsample2(x)), K = 3
POLY = _env.get("NEF_POLY", "1") == "1"
_POLY_FWD = _env.get("NEF_POLY_FWD", "1") == "1"
_POLY_W = _env.get("NEF_POLY_W", "1") == "1"


def poly_weights(w, tile_Cr=0):
    """w [R, Cig, 3] -> [2R, Cig, 3]: row 2r + p = the phase-p weights of row r (csrc/elementwise.hip poly_weights_kernel): output
    2m + p of conv1d(upsample2(x), w) is the K = 3 conv of the half-resolution x with them.  `tile_Cr` (channels per group): rows in
    the tile order the polyphase FORWARD launch wants instead."""
    _chk(w)
    ws = torch.empty(2 * w.shape[0], w.shape[1], 3, device=w.device, dtype=torch.float32)
    _lib.check(_lib.load().nef_poly_weights(_p(w), _p(ws), w.shape[0], w.shape[1], int(tile_Cr), _stream()), "nef_poly_weights")
    return ws


def poly_fwd_ok(G, Cog, Cig, T):
    """Can y = conv1d(upsample2(x [.., G*Cig, T/2]), w [G*Cog, Cig, 3]) run in polyphase form?  (a split-fp16 conv with Cig reduction
    channels, 2 Cog output rows in 128-row tiles, T / 2 columns)"""
    Th = T // 2
    return (POLY and _POLY_FWD and H2 and T % 2 == 0 and Th >= 128 and Cog % 64 == 0 and Cig % 16 == 0 and h2_ok(3, Cig, 2 * Cog, Th, 1) and
            _H2_DIR[False] and "3" in _H2_K and _h2_fills(G, 2 * Cog, Th, 256, 128))


def conv_poly_fwd(xv, w, Cog, bias=None, pro=None, stats=False, site=None, save_edge=False):
    """y [B, G*Cog, 2T] = conv1d(upsample2(prologue(x)), w) + bias in polyphase form: one split-fp16 conv over the half-resolution
    x (`xv`, a GV) whose rows are the two phases of each channel (conv args pro_mode 8 | affine bit), written interleaved, + the two
    row-end columns (nef_poly_fwd_edge).  `pro` = (mode, a, b, Bp) as conv() gets it for this layer (bit1 = the upsampling itself,
    bit0 the BatchNorm affine + ReLU of the layer below); `stats`: also leave the BatchNorm slot sums -> (y, (slots, nslot)).
    Reference: codes/network/model_nefnet.py:102-105 (nn.Upsample + the DoubleConv's first conv).  Summation order differs from
    the upsampling-prologue form (fp32-class either way)."""
    L = _lib.load()
    G, Cig, Th = xv.G, xv.Cg, xv.T
    assert w.shape == (G * Cog, Cig, 3) and xv.bs == G * Cig * Th and xv.gs == Cig * Th, "conv_poly_fwd: dense input rows"
    aff = pro is not None and bool(pro[0] & 1)
    # the phase weights are formed inside the pack (nef_pack_desc.src_mode 1): no poly_weights launch, no fp32 phase tensor
    wp = pack_weight(w, G, T=Th, site=w.data_ptr() if site is None else site, plain=False, src=("poly", Cog))
    if int(getattr(wp, "nef_wino", 0)) != 3:
        raise _lib.NefLibraryError("conv_poly_fwd: shape outside the split-fp16 kernel (ask poly_fwd_ok first)")
    y = torch.empty(xv.B, G * Cog, 2 * Th, device=xv.t.device, dtype=torch.float32)
    slots = conv_stats_buffer(wp, xv.B, G, Cog, Th, xv.t.device) if stats else None
    out = GV(y, xv.B, G, 2 * Cog, Th, G * Cog * 2 * Th, Cog * 2 * Th, 0)      # (strides of the real tensor; Cg / T as the launch counts them)
    conv(xv, wp, 2 * Cog, 3, out=out, bias=bias, pro=(8 | int(aff), pro[1] if aff else None, pro[2] if aff else None, pro[3] if aff else 1),
         stats=slots)
    pa, pb, pbp = (_p(pro[1]), _p(pro[2]), pro[3]) if aff else (None, None, 1)
    xedge = torch.empty(xv.B, G * Cig, 2, device=xv.t.device, dtype=torch.float32) if save_edge else None
    _lib.check(L.nef_poly_fwd_edge(xv.ptr, _p(w), _p(y), xv.B, G, Cog, Cig, 2 * Th, pa, pb, pbp,
                                   _p(slots[0]) if slots is not None else None, slots[1] if slots is not None else 0, _p(xedge),
                                   _stream()), "nef_poly_fwd_edge")
    y.nef_xedge = xedge      # (the prologue's output at both row ends: what the polyphase weight gradient's row-end terms need)
    return (y, slots) if stats else y


def poly_bwd_ok(G, Cog, Cig, T):
    """Can the backward-data pass of conv1d(upsample2(x) [.., G*Cig, T], w [G*Cog, Cig, 3]) run in polyphase form?  (the launch is a
    split-fp16 conv with 2 Cog reduction channels, Cig outputs, T / 2 columns)"""
    Th = T // 2
    return (POLY and H2 and T % 2 == 0 and Th >= 128 and Cig % 64 == 0 and (2 * Cog) % 16 == 0 and h2_ok(3, 2 * Cog, Cig, Th) and
            _H2_DIR[True] and "3" in _H2_K and _h2_fills(G, Cig, Th, 256, 128 if Cig % 128 == 0 else 64))


def conv_bwd_data_poly(gyv, w, Cig, bnb=None, site=None, phase_major=False):
    """Gradient wrt the HALF-resolution input x of y = conv1d(upsample2(x), w) (K = 3, zero padding 1), from gy [B, G*Cog, T]
    (`gyv`, a GV): one split-fp16 conv at half resolution over the 2 Cog phase channels of gy (conv args pro_mode 4) + the row-end
    terms (nef_poly_bwd_edge) -- the full-resolution gradient wrt upsample2(x) is never formed.  `w` [G*Cog, Cig, 3] is the conv's
    own weight; `bnb` as in conv() (plain form: its x at the half resolution).  Reference semantics: autograd through
    codes/network/model_nefnet.py:102-105.  Summation order differs from conv + upsample2_bwd (fp32-class either way)."""
    L = _lib.load()
    if phase_major:      # `gyv`: the gradient as [B, G * 2 Cog, T/2] (bn_relu_bwd(..., phase_major=True)): a plain conv over it
        G, Cog, T = gyv.G, gyv.Cg // 2, 2 * gyv.T
    else:
        G, Cog, T = gyv.G, gyv.Cg, gyv.T
    Th = T // 2
    assert w.shape == (G * Cog, Cig, 3) and gyv.gs == Cog * T, "conv_bwd_data_poly: dense gradient rows"
    wp = pack_weight(w, G, flip=True, T=Th, site=w.data_ptr() if site is None else site, src=("poly", 0))
    if int(getattr(wp, "nef_wino", 0)) != 3:
        raise _lib.NefLibraryError("conv_bwd_data_poly: shape outside the split-fp16 kernel (ask poly_bwd_ok first)")
    xv = GV(gyv.t, gyv.B, G, 2 * Cog, Th, gyv.bs, gyv.gs, gyv.off)
    if bnb is not None and len(bnb) == 6:      # (x, mean, invstd, a, b, Bp): the slot buffer is made here (returned as g.nef_slots)
        slots = conv_stats_buffer(wp, gyv.B, G, Cig, Th, gyv.t.device)
        bnb = None if slots is None else (*bnb, slots)
    g = conv(xv, wp, Cig, 3, role="conv_bwd_data", bnb=bnb, pro=None if phase_major else (4, None, None, 1),
             tag_extra="pm" if phase_major else "")      # (bench.py: "pm" = plain conv over the phase-major gradient)
    g.nef_slots = bnb[6] if bnb is not None else None
    ba = [None] * 5 + [1, None, 0]
    if bnb is not None:
        ba = [_p(t) for t in bnb[:5]] + [bnb[5], _p(bnb[6][0]), bnb[6][1]]
    _lib.check(L.nef_poly_bwd_edge(gyv.ptr, _p(w), _p(g), gyv.B, G, Cog, Cig, T, *ba, int(phase_major), _stream()), "nef_poly_bwd_edge")
    return g


def poly_w_ok(B, G, Cog, Cig, T):
    """Weight gradient of conv1d(upsample2(x), w) in polyphase form: a split-fp16 weight gradient over the half-resolution x and
    the phase-major gradient (2 Cog rows), producer / consumer form (2 Cog % 128 == 0), then nef_poly_wgrad_fold."""
    Th = T // 2
    return (POLY and _POLY_W and T % 4 == 0 and Cog % 64 == 0 and h2w_ok(3, Cig, 2 * Cog, Th, 1) and
            (BATCH_HINT is None or B * ((Th + 63) // 64) >= 8 * _H2_MIN_WGS))


def conv_bwd_weight_poly(xv, gy_pm, Cog, pro, xedge, site=None):
    """gw [G*Cog, Cig, 3] of y = conv1d(upsample2(prologue(x)), w) from the half-resolution x (`xv`), the PHASE-MAJOR gradient
    gy_pm [B, G*2Cog, T/2] and `xedge` [B, G*Cig, 2] (the prologue's output at both row ends, conv_poly_fwd(..., save_edge=True)).
    `pro` as the forward got it (bit0 = affine + ReLU prologue; the upsampling bit is what the polyphase form replaces)."""
    L = _lib.load()
    G, Cig = xv.G, xv.Cg
    aff = pro is not None and bool(pro[0] & 1)
    gyv = GV.dense(gy_pm, G)
    gw2 = conv_bwd_weight(xv, gyv, 3, pro=(4 | int(aff), pro[1] if aff else None, pro[2] if aff else None, pro[3] if aff else 1),
                          site=site, h2=True)
    gw = torch.empty(G * Cog, Cig, 3, device=gy_pm.device, dtype=torch.float32)
    n = L.nef_poly_wgrad_fold_ws_bytes(xv.B, G, Cog, Cig)
    ws = torch.empty(n // 4, device=gy_pm.device, dtype=torch.float32)
    _lib.check(L.nef_poly_wgrad_fold(_p(gw2), _p(gy_pm), _p(xedge), _p(gw), _p(ws), n, xv.B, G, Cog, Cig, 2 * xv.T, _stream()),
               "nef_poly_wgrad_fold")
    return gw


def h2w_ok(K, Cig, Cog, T, pro_mode=0, in_scale=False):
    return (H2 and _H2_W and str(K) in _H2_WK and T % 2 == 0 and T >= max(64, _H2_MIN_T) and Cig % 64 == 0 and Cog % 64 == 0 and
            (K == 3 or not pro_mode) and not (pro_mode and in_scale))


def conv_bwd_weight(xv, gyv, K, in_scale=None, pro=None, wino=None, site=None, h2=None, x_scale=0.0, gy_scale=0.0):
    """gw [G*Cog, Cig, K] for y = conv(prologue(x) * in_scale, w); `pro` as in conv().  Default path: the split-fp16 kernel
    (csrc/conv_h2w.hip) wherever its shape rules hold (h2w_ok; `h2=False` forbids it, giving `wino` forbids it too); `site`: the
    identity of the call site for its operand-magnitude slots (the engine passes the weight's address; None = measure at every
    call), `x_scale` / `gy_scale`: explicit powers of two instead.  `wino`: force (True / 4) or forbid
    (False) the transposed-Winograd form -- F(3,4) for K == 3, the 4 + 3 split through F(4,4) + F(3,4) for K == 7; default:
    wherever it applies (see WINOGRAD, WINO_BW4, WINO_BW7)."""
    L = _lib.load()
    B, T, G, Cig, Cog = xv.B, gyv.T, xv.G, xv.Cg, gyv.Cg
    gw = torch.empty(G * Cog, Cig, K, device=xv.t.device, dtype=torch.float32)
    pm0 = pro[0] if pro is not None else 0
    if h2 is None:
        h2 = (wino is None and h2w_ok(K, Cig, Cog, T, pm0, in_scale is not None) and
              (BATCH_HINT is None or B * ((T + 63) // 64) >= 8 * _H2_MIN_WGS))      # enough (sample, tile) steps to split
    if h2:
        n = L.nef_conv_bwd_weight_h2_ws_bytes(B, T, G, Cig, Cog, K)
        if n == 0:
            raise _lib.NefLibraryError(f"conv_bwd_weight (split-fp16): unsupported shape Cig={Cig} Cog={Cog} K={K} T={T}")
        ws = workspace(n, xv.t.device)
        sc, sc_bs, sc_gs = (None, 0, 0) if in_scale is None else (_p(in_scale[0]), in_scale[1], in_scale[2])
        pm, pa, pb, pbp = (pro[0], _p(pro[1]), _p(pro[2]), pro[3]) if pm0 else (0, None, None, 1)

        def launch(amax, nxt, clamped=None):
            _lib.check(L.nef_conv_bwd_weight_h2(xv.ptr, xv.bs, xv.gs, sc, sc_bs, sc_gs, pa, pb, pm, pbp, gyv.ptr, gyv.bs, gyv.gs,
                                                _p(gw), _p(ws), n, B, T, G, Cig, Cog, K, float(x_scale), float(gy_scale),
                                                amax, None if amax is None else amax + 4, nxt, None if nxt is None else nxt + 4,
                                                clamped, _stream()), "nef_conv_bwd_weight_h2")
        tag = ("conv_bwd_weight", K, G, Cig, Cog, B, T) + (("up",) if pm0 & 2 else (("pw",) if pm0 & 4 else ()))
        ev = _timed(tag)
        if ev is not None:
            EXEC_FRAC[tag] = 0.0
            EXEC_FP16[tag] = 3.0
        if x_scale and gy_scale:
            launch(None, None)
        else:
            st = _amax_state(xv.t.device)
            key = None
            if AMAX_SCOPE is not None and site is not None and _H2_AMAX != "anon":
                occ = st["occ"]
                base = (AMAX_SCOPE, (site, "w"), "conv_bwd_weight", B, T)
                k = occ[base] = occ.get(base, 0) + 1
                key = base + (k,)
            i = _amax_index(st, key if key is not None else (None, "bww"), 2)          # slots i (x) and i + 1 (gy); unscoped launches: a pair of their own
            nxt = st["nxt"].data_ptr() + 4 * i
            if i not in st["ready"] or key is None:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("split-fp16 weight gradient: a call site's first launch cannot be captured")
                st["nxt"][i:i + 2] = 0.0
                launch(None, nxt)
                st["cur"][i:i + 2] = st["nxt"][i:i + 2]
                if key is not None:
                    _note_tail(st, i, 2, xv.t, gyv.t)
                st["ready"].add(i)
            st["used"] = True
            launch(st["cur"].data_ptr() + 4 * i, nxt, st["clamped"].data_ptr())
        if ev is not None:
            ev.record()
        return gw
    n = L.nef_conv_bwd_weight_ws_bytes(B, T, G, Cig, Cog, K)
    if n == 0:
        raise _lib.NefLibraryError(f"conv_bwd_weight: unsupported shape Cig={Cig} Cog={Cog} K={K}")
    ws = workspace(n, xv.t.device)
    sc, sc_bs, sc_gs = (None, 0, 0) if in_scale is None else (_p(in_scale[0]), in_scale[1], in_scale[2])
    tag = ("conv_bwd_weight", K, G, Cig, Cog, B, T) + (("up",) if pm0 & 2 else ())
    ev = _timed(tag)
    if wino is None:
        wino = (WINOGRAD and ((K == 3 and WINO_BW4) or (K == 7 and WINO_BW7)) and T % 2 == 0 and T >= 64
                and not (K == 7 and pro is not None and pro[0]))
    wino = 4 if wino else False
    if ev is not None:
        EXEC_FRAC[tag] = _exec_frac(K, 2) if wino else 1.0
        EXEC_FP16[tag] = 0.0
    if wino:
        pm, pa, pb, pbp = (pro[0], _p(pro[1]), _p(pro[2]), pro[3]) if (pro is not None and pro[0]) else (0, None, None, 1)
        _lib.check(L.nef_conv_bwd_weight_wino4(xv.ptr, xv.bs, xv.gs, sc, sc_bs, sc_gs, pa, pb, pm, pbp, gyv.ptr, gyv.bs,
                                              gyv.gs, _p(gw), _p(ws), n, B, T, G, Cig, Cog, K, _stream()),
                   "nef_conv_bwd_weight_wino4")
    elif pro is not None and pro[0]:
        assert in_scale is None
        _lib.check(L.nef_conv_bwd_weight_pro(xv.ptr, xv.bs, xv.gs, _p(pro[1]), _p(pro[2]), pro[0], pro[3], gyv.ptr, gyv.bs,
                                             gyv.gs, _p(gw), _p(ws), n, B, T, G, Cig, Cog, K, _stream()),
                   "nef_conv_bwd_weight_pro")
    else:
        _lib.check(L.nef_conv_bwd_weight(xv.ptr, xv.bs, xv.gs, sc, sc_bs, sc_gs, gyv.ptr, gyv.bs, gyv.gs, _p(gw), _p(ws),
                                         n, B, T, G, Cig, Cog, K, _stream()), "nef_conv_bwd_weight")
    if ev is not None:
        ev.record()
    return gw


def chan_sum(x):
    L = _lib.load()
    _chk(x)
    B, Ct, T = x.shape
    out = torch.empty(Ct, device=x.device, dtype=torch.float32)
    n = L.nef_chan_sum_ws_bytes(Ct)
    ws = workspace(n, x.device)
    _lib.check(L.nef_chan_sum(_p(x), _p(out), _p(ws), n, B, Ct, T, _stream()), "nef_chan_sum")
    return out


# ------------------------------------------------------------------ transposed conv (k2 s2)
def _convt2_fma_fwd(x, w, bias, G):
    L = _lib.load()
    _chk(x), _chk(w), _chk(bias)
    B, Ct, T = x.shape
    Cig, Cog = Ct // G, w.shape[1]
    y = torch.empty(B, G * Cog, 2 * T, device=x.device, dtype=torch.float32)
    _lib.check(L.nef_convt2_fwd(_p(x), _p(w), _p(bias), _p(y), B, G, Cig, Cog, T, _stream()), "nef_convt2_fwd")
    return y


def _convt2_fma_bwd_data(gy, w, G):
    L = _lib.load()
    _chk(gy), _chk(w)
    B, Ct, To = gy.shape
    Cog, Cig, T = Ct // G, w.shape[0] // G, To // 2
    gx = torch.empty(B, G * Cig, T, device=gy.device, dtype=torch.float32)
    _lib.check(L.nef_convt2_bwd_data(_p(gy), _p(w), _p(gx), B, G, Cig, Cog, T, _stream()), "nef_convt2_bwd_data")
    return gx


def _convt2_fma_bwd_weight(x, gy, G):
    L = _lib.load()
    _chk(x), _chk(gy)
    B, Ct, T = x.shape
    Cig, Cog = Ct // G, gy.shape[1] // G
    gw = torch.empty(G * Cig, Cog, 2, device=x.device, dtype=torch.float32)
    gb = torch.empty(G * Cog, device=x.device, dtype=torch.float32)
    n = L.nef_convt2_bwd_weight_ws_bytes(G, Cig, Cog)
    ws = workspace(n, x.device)
    _lib.check(L.nef_convt2_bwd_weight(_p(x), _p(gy), _p(gw), _p(gb), _p(ws), n, B, G, Cig, Cog, T, _stream()),
               "nef_convt2_bwd_weight")
    return gw, gb


def group_transpose(t, G, R, Cn):
    """out[g][c][r] = in[g][r][c] over a flat [G*R*Cn] fp32 buffer."""
    L = _lib.load()
    _chk(t)
    out = torch.empty_like(t)
    _lib.check(L.nef_group_transpose(_p(t), _p(out), G, R, Cn, _stream()), "nef_group_transpose")
    return out


def convt2_as_conv_weight(w, G):
    """ConvTranspose1d weight [G*Cig, Cog, 2] -> the 1x1 conv weight [G*2Cog, Cig, 1] onto channels m = co*2+j."""
    Cig, Cog = w.shape[0] // G, w.shape[1]
    return group_transpose(w, G, Cig, 2 * Cog).view(G * 2 * Cog, Cig, 1)


def convt2_deinterleave(gy):
    L = _lib.load()
    _chk(gy)
    B, Ct, To = gy.shape
    gyq = torch.empty(B, 2 * Ct, To // 2, device=gy.device, dtype=torch.float32)
    _lib.check(L.nef_convt2_deinterleave(_p(gy), _p(gyq), B, Ct, To // 2, _stream()), "nef_convt2_deinterleave")
    return gyq


def convt2_fwd(x, w, bias, G, fma=False):
    """ConvTranspose1d(k=2, s=2, groups=G, bias).  Default: grouped 1x1 conv on the matrix cores + interleave;
    `fma=True` selects the plain-FMA kernel (kept as a cross-check)."""
    if fma:
        return _convt2_fma_fwd(x, w, bias, G)
    L = _lib.load()
    _chk(x), _chk(w), _chk(bias)
    B, Ct, T = x.shape
    Cog = w.shape[1]
    yq = conv(GV.dense(x, G), pack_weight(convt2_as_conv_weight(w, G), G), 2 * Cog, 1)     # [B, G*2Cog, T]
    y = torch.empty(B, G * Cog, 2 * T, device=x.device, dtype=torch.float32)
    _lib.check(L.nef_convt2_interleave(_p(yq), _p(bias), _p(y), B, G * Cog, T, _stream()), "nef_convt2_interleave")
    return y


def convt2_bwd_data(gy, w, G, fma=False, gyq=None):
    if fma:
        return _convt2_fma_bwd_data(gy, w, G)
    Cig = w.shape[0] // G
    gyq = convt2_deinterleave(gy) if gyq is None else gyq
    return conv(GV.dense(gyq, G), pack_weight(convt2_as_conv_weight(w, G), G, flip=True), Cig, 1, role="conv_bwd_data")


def convt2_bwd_weight(x, gy, G, fma=False, gyq=None):
    if fma:
        return _convt2_fma_bwd_weight(x, gy, G)
    B, Ct, T = x.shape
    Cig, Cog = Ct // G, gy.shape[1] // G
    gyq = convt2_deinterleave(gy) if gyq is None else gyq
    gwc = conv_bwd_weight(GV.dense(x, G), GV.dense(gyq, G), 1)                              # [G*2Cog, Cig, 1]
    gw = group_transpose(gwc.contiguous(), G, 2 * Cog, Cig).view(G * Cig, Cog, 2)
    return gw, chan_sum(gy)


# ------------------------------------------------------------------ angular encoding + MLP
def theta_mlp_fwd(theta, W, bias):
    L = _lib.load()
    _chk(theta), _chk(W), _chk(bias)
    N, O = theta.numel() // 2, W.shape[0]
    y = torch.empty(*theta.shape[:-1], O, device=theta.device, dtype=torch.float32)
    _lib.check(L.nef_theta_mlp_fwd(_p(theta), _p(W), _p(bias), _p(y), N, O, _stream()), "nef_theta_mlp_fwd")
    return y


def theta_mlp_bwd(theta, gy, O):
    L = _lib.load()
    _chk(theta), _chk(gy)
    N = theta.numel() // 2
    gW = torch.empty(O, 12, device=theta.device, dtype=torch.float32)
    gb = torch.empty(O, device=theta.device, dtype=torch.float32)
    _lib.check(L.nef_theta_mlp_bwd(_p(theta), _p(gy), _p(gW), _p(gb), N, O, _stream()), "nef_theta_mlp_bwd")
    return gW, gb


def theta_encode(theta):
    L = _lib.load()
    _chk(theta)
    N = theta.numel() // 2
    enc = torch.empty(*theta.shape[:-1], 12, device=theta.device, dtype=torch.float32)
    _lib.check(L.nef_theta_encode(_p(theta), _p(enc), N, _stream()), "nef_theta_encode")
    return enc


# ------------------------------------------------------------------ elementwise
def chscale_fwd(x, s, s_bs=None):
    L = _lib.load()
    _chk(x)
    B, Ct, T = x.shape
    y = torch.empty_like(x)
    ev = _hbm("chscale_fwd", x, y)
    _lib.check(L.nef_chscale_fwd(_p(x), _p(s), Ct if s_bs is None else s_bs, _p(y), B, Ct, T, _stream()), "nef_chscale_fwd")
    _done(ev)
    return y


def slots_to_rows(slots, B):
    """[B, C] per-(sample, channel) sums of the word-0 slot values a conv epilogue left (conv(..., stats=..., stats_mode=1))."""
    sl, nslot = slots
    out = torch.empty(B, sl.shape[0], device=sl.device, dtype=torch.float32)
    _lib.check(_lib.load().nef_slots_to_rows(_p(sl), nslot, _p(out), B, sl.shape[0], _stream()), "nef_slots_to_rows")
    return out


def chscale_bwd(gy, x, s, s_bs=None, relu_x=False):
    """`relu_x`: x is a ReLU output -- gx also carries that ReLU's backward mask (x > 0)."""
    L = _lib.load()
    _chk(gy), _chk(x)
    B, Ct, T = x.shape
    gx = torch.empty_like(x)
    gs = torch.empty(B, Ct, device=x.device, dtype=torch.float32)
    ev = _hbm("chscale_bwd", gy, x, gx)
    _lib.check(L.nef_chscale_bwd(_p(gy), _p(x), _p(s), Ct if s_bs is None else s_bs, _p(gx), _p(gs), B, Ct, T,
                                 int(relu_x), _stream()),
               "nef_chscale_bwd")
    _done(ev)
    return gx, gs


def gate(g, ref, scale=1.0):
    L = _lib.load()
    _chk(g), _chk(ref)
    out = torch.empty_like(g)
    _lib.check(L.nef_gate(_p(g), _p(ref), _p(out), scale, g.numel(), _stream()), "nef_gate")
    return out


def add(a, b):
    L = _lib.load()
    _chk(a), _chk(b)
    out = torch.empty_like(a)
    _lib.check(L.nef_add(_p(a), _p(b), _p(out), a.numel(), _stream()), "nef_add")
    return out


# ------------------------------------------------------------------ ROI ops
def roi_align_fwd(z, rois, T=None, t_off=0):
    """z [B,C,T], or a time window of it [B,C,zT] that starts at `t_off` of a length-`T` axis."""
    L = _lib.load()
    _chk(z), _chk(rois, torch.int64)
    B, Ct, zT = z.shape
    T = zT if T is None else T
    out = torch.empty(B, Ct, N_SEG, ROI_BINS, device=z.device, dtype=torch.float32)
    _lib.check(L.nef_roi_align_fwd(_p(z), _p(rois), _p(out), B, Ct, T, zT, t_off, _stream()), "nef_roi_align_fwd")
    return out


def roi_align_bwd(gout, rois, T, zT=None, t_off=0):
    """Gradient wrt z; with (zT, t_off) only that time window is produced (everything outside it is zero)."""
    L = _lib.load()
    _chk(gout), _chk(rois, torch.int64)
    B, Ct = gout.shape[0], gout.shape[1]
    zT = T if zT is None else zT
    gz = torch.empty(B, Ct, zT, device=gout.device, dtype=torch.float32)
    _lib.check(L.nef_roi_align_bwd(_p(gout), _p(rois), _p(gz), B, Ct, T, zT, t_off, _stream()), "nef_roi_align_bwd")
    return gz


def window_crop(xv, t0, W):
    """Dense [B, G*Cg, W] copy of the time window [t0, t0+W) of a grouped view."""
    L = _lib.load()
    dst = torch.empty(xv.B, xv.G * xv.Cg, W, device=xv.t.device, dtype=torch.float32)
    _lib.check(L.nef_window_crop(xv.ptr, xv.bs, xv.gs, _p(dst), xv.B, xv.G, xv.Cg, xv.T, t0, W, _stream()), "nef_window_crop")
    return dst


def window_scatter(src, outv, t0):
    """Write `src` [B, G*Cg, W] into the window [t0, t0+W) of the grouped view `outv`, zero everywhere else."""
    L = _lib.load()
    _chk(src)
    _lib.check(L.nef_window_scatter(_p(src), outv.ptr, outv.bs, outv.gs, outv.B, outv.G, outv.Cg, outv.T, t0,
                                    src.shape[2], _stream()), "nef_window_scatter")


def roi_unpool_fwd(zseg, rois, T, status=None):
    L = _lib.load()
    _chk(zseg), _chk(rois, torch.int64)
    B, Ct = zseg.shape[0], zseg.shape[1]
    out = torch.empty(B, Ct, T, device=zseg.device, dtype=torch.float32)
    ev = _hbm("roi_unpool_fwd", zseg, out)
    _lib.check(L.nef_roi_unpool_fwd(_p(zseg), _p(rois), _p(out), _p(status), B, Ct, T, _stream()), "nef_roi_unpool_fwd")
    _done(ev)
    return out


def roi_unpool_bwd(gout, rois):
    L = _lib.load()
    _chk(gout), _chk(rois, torch.int64)
    B, Ct, T = gout.shape
    gz = torch.empty(B, Ct, N_SEG, 2 * ROI_BINS, device=gout.device, dtype=torch.float32)
    ev = _hbm("roi_unpool_bwd", gout, gz)
    _lib.check(L.nef_roi_unpool_bwd(_p(gout), _p(rois), _p(gz), B, Ct, T, _stream()), "nef_roi_unpool_bwd")
    _done(ev)
    return gz


def roi_segment_table(rois):
    L = _lib.load()
    _chk(rois, torch.int64)
    B = rois.shape[0]
    start = torch.empty(B, N_SEG, device=rois.device, dtype=torch.int64)
    length = torch.empty(B, N_SEG, device=rois.device, dtype=torch.int64)
    _lib.check(L.nef_roi_segment_table(_p(rois), _p(start), _p(length), B, _stream()), "nef_roi_segment_table")
    return start, length


# ------------------------------------------------------------------ latent mix
def lead_mean(z1, z2r, V):
    L = _lib.load()
    _chk(z1), _chk(z2r)
    B, _, T = z1.shape
    latent = torch.empty(B, 256, T, device=z1.device, dtype=torch.float32)
    ev = _hbm("lead_mean", z1, z2r, latent)
    _lib.check(L.nef_lead_mean(_p(z1), _p(z2r), _p(latent), B, V, T, _stream()), "nef_lead_mean")
    _done(ev)
    return latent


def _choice(c):
    """(c1, c2) as ints, or a device int32[2] tensor (graph replay) -> (c1, c2, device pointer)."""
    if torch.is_tensor(c):
        return 0, 0, c.data_ptr()
    return int(c[0]), int(c[1]), None


def mix_fwd(latent, z1, z2r, q, V, c1, c2=None):
    L = _lib.load()
    c1, c2, cdev = _choice(c1 if c2 is None else (c1, c2))
    _chk(latent), _chk(q)
    B, _, T = latent.shape
    D = torch.empty(3 * B, 256, T, device=latent.device, dtype=torch.float32)
    _lib.check(L.nef_mix_fwd(_p(latent), _p(z1), _p(z2r), _p(q), _p(D), B, V, T, c1, c2, cdev, _stream()),
               "nef_mix_fwd")
    return D


def mix_fwd_shared(latent, z1, z2r, q, V, c1, c2=None):
    """The two DISTINCT decoder inputs, [2B,256,T] = (q*cat(z1m|z2m) | q*cat(z1[c1]|z2r[c2]))."""
    L = _lib.load()
    c1, c2, cdev = _choice(c1 if c2 is None else (c1, c2))
    _chk(latent), _chk(q)
    B, _, T = latent.shape
    D2 = torch.empty(2 * B, 256, T, device=latent.device, dtype=torch.float32)
    ev = _hbm("mix_fwd_shared", latent, z1, z2r, D2)
    _lib.check(L.nef_mix_fwd_shared(_p(latent), _p(z1), _p(z2r), _p(q), _p(D2), B, V, T, c1, c2, cdev, _stream()),
               "nef_mix_fwd_shared")
    _done(ev)
    return D2


def lead_mean_mix_shared(z1, z2r, q, V, c1, c2=None):
    """lead_mean + mix_fwd_shared in one pass: (latent [B,256,T], D2 [2B,256,T]), bit-identical to the two calls."""
    L = _lib.load()
    c1, c2, cdev = _choice(c1 if c2 is None else (c1, c2))
    _chk(z1), _chk(z2r), _chk(q)
    B, _, T = z1.shape
    latent = torch.empty(B, 256, T, device=z1.device, dtype=torch.float32)
    D2 = torch.empty(2 * B, 256, T, device=z1.device, dtype=torch.float32)
    ev = _hbm("lead_mean_mix_shared", z1, z2r, latent, D2)
    _lib.check(L.nef_lead_mean_mix_shared(_p(z1), _p(z2r), _p(q), _p(latent), _p(D2), B, V, T, c1, c2, cdev, _stream()),
               "nef_lead_mean_mix_shared")
    _done(ev)
    return latent, D2


def mix_bwd_shared_up(gU2, latent, z1, z2r, q, V, c1, c2=None, relu_z1=False):
    """mix_bwd for the two-pass gradient wrt the shared input: gU2 [2B,256,2T] wrt its x2-upsampled form (the adjoint is taken on
    the fly), or [2B,256,T] wrt the input itself (what the polyphase backward-data pass leaves)."""
    L = _lib.load()
    c1, c2, cdev = _choice(c1 if c2 is None else (c1, c2))
    _chk(gU2)
    B, _, T = latent.shape
    up = gU2.shape[2] == 2 * T
    assert gU2.shape == (2 * B, 256, 2 * T if up else T)
    gz1, gz2r = torch.empty_like(z1), torch.empty_like(z2r)
    gq = torch.empty(B, 256, device=latent.device, dtype=torch.float32)
    name = "mix_bwd_shared_up" if up else "mix_bwd_shared"
    ev = _hbm(name, gU2, z1, z2r, gz1, gz2r)
    fn = L.nef_mix_bwd_shared_up if up else L.nef_mix_bwd_shared
    _lib.check(fn(_p(gU2), _p(latent), _p(z1), _p(z2r), _p(q), _p(gz1), _p(gz2r), _p(gq), B, V, T, c1, c2, cdev, int(relu_z1),
                  _stream()), "nef_" + name)
    _done(ev)
    return gz1, gz2r, gq


def pass_combine_fwd(P2, bias, B):
    """P2 [2B,2C,L] (A | B half-conv outputs of the mean / picked inputs) -> c1 [3B,C,L] of the three Standin passes."""
    L = _lib.load()
    _chk(P2), _chk(bias)
    C2, Ln = P2.shape[1], P2.shape[2]
    c1 = torch.empty(3 * B, C2 // 2, Ln, device=P2.device, dtype=torch.float32)
    ev = _hbm("pass_combine_fwd", P2, c1)
    _lib.check(L.nef_pass_combine_fwd(_p(P2), _p(bias), _p(c1), B, C2 // 2, Ln, _stream()), "nef_pass_combine_fwd")
    _done(ev)
    return c1


def pass_combine_fwd_stats(P2, bias, B, gamma, beta, running_mean, running_var, eps=1e-5, momentum=0.1, nbt=None):
    """pass_combine_fwd + the train-mode BatchNorm statistics of its output: returns (c1, mean, invstd, a, b)."""
    L = _lib.load()
    _chk(P2), _chk(bias)
    C2, Ln = P2.shape[1], P2.shape[2]
    Ct = C2 // 2
    c1 = torch.empty(3 * B, Ct, Ln, device=P2.device, dtype=torch.float32)
    mean, invstd, a, b = (torch.empty(3, Ct, device=P2.device, dtype=torch.float32) for _ in range(4))
    n = L.nef_pass_combine_stats_ws_bytes(B, Ct)
    ws = workspace(n, P2.device)
    ev = _hbm("pass_combine_fwd", P2, c1)
    _lib.check(L.nef_pass_combine_fwd_stats(_p(P2), _p(bias), _p(c1), _p(gamma), _p(beta), _p(running_mean),
                                            _p(running_var), _p(mean), _p(invstd), _p(a), _p(b), _p(ws), n, B, Ct, Ln, eps,
                                            momentum, _p(nbt), _stream()), "nef_pass_combine_fwd_stats")
    _done(ev)
    return c1, mean, invstd, a, b


def pass_combine_bwd(gc1):
    L = _lib.load()
    _chk(gc1)
    B3, Ct, Ln = gc1.shape
    gP2 = torch.empty(2 * (B3 // 3), 2 * Ct, Ln, device=gc1.device, dtype=torch.float32)
    _lib.check(L.nef_pass_combine_bwd(_p(gc1), _p(gP2), B3 // 3, Ct, Ln, _stream()), "nef_pass_combine_bwd")
    return gP2


def mix_bwd(gD, latent, z1, z2r, q, V, c1, c2=None, upsampled=False, relu_z1=False):
    """`upsampled`: gD is the gradient wrt the x2-upsampled decoder input [3B,256,2T]; its adjoint is taken on the fly."""
    L = _lib.load()
    c1, c2, cdev = _choice(c1 if c2 is None else (c1, c2))
    _chk(gD)
    B, _, T = latent.shape
    assert gD.shape[2] == (2 * T if upsampled else T)
    gz1, gz2r = torch.empty_like(z1), torch.empty_like(z2r)
    gq = torch.empty(B, 256, device=latent.device, dtype=torch.float32)
    fn = L.nef_mix_bwd_up if upsampled else L.nef_mix_bwd
    # relu_z1: z1 is a ReLU output; gz1 also carries that ReLU's backward mask
    _lib.check(fn(_p(gD), _p(latent), _p(z1), _p(z2r), _p(q), _p(gz1), _p(gz2r), _p(gq), B, V, T, c1, c2, cdev,
                  int(relu_z1), _stream()),
               "nef_mix_bwd")
    return gz1, gz2r, gq


# ------------------------------------------------------------------ decoder pieces
def upsample2_fwd(x):
    L = _lib.load()
    _chk(x)
    N, Ct, T = x.shape
    y = torch.empty(N, Ct, 2 * T, device=x.device, dtype=torch.float32)
    _lib.check(L.nef_upsample2_fwd(_p(x), _p(y), N * Ct, T, _stream()), "nef_upsample2_fwd")
    return y


def upsample2_aff_fwd(x, a, b, Bp):
    """upsample2(relu(x*a[p,c] + b[p,c])), p = sample // Bp."""
    L = _lib.load()
    _chk(x)
    N, Ct, T = x.shape
    y = torch.empty(N, Ct, 2 * T, device=x.device, dtype=torch.float32)
    ev = _hbm("upsample2_aff_fwd", x, y)
    _lib.check(L.nef_upsample2_aff_fwd(_p(x), _p(a), _p(b), _p(y), N, Ct, T, Bp, _stream()), "nef_upsample2_aff_fwd")
    _done(ev)
    return y


def upsample2_bwd(gy):
    L = _lib.load()
    _chk(gy)
    N, Ct, To = gy.shape
    gx = torch.empty(N, Ct, To // 2, device=gy.device, dtype=torch.float32)
    _lib.check(L.nef_upsample2_bwd(_p(gy), _p(gx), N * Ct, To // 2, _stream()), "nef_upsample2_bwd")
    return gx


def bn_train_stats(x, gamma, beta, running_mean, running_var, P, eps=1e-5, momentum=0.1, nbt=None):
    """Returns (mean, invstd, a, b), each [P, C]; updates the running statistics in place, pass by pass."""
    L = _lib.load()
    _chk(x)
    N, Ct, Ln = x.shape
    Bp = N // P
    mean, invstd, a, b = (torch.empty(P, Ct, device=x.device, dtype=torch.float32) for _ in range(4))
    n = L.nef_bn_ws_bytes(P, Ct)
    ws = workspace(n, x.device)
    ev = _hbm("bn_train_stats", x)
    _lib.check(L.nef_bn_train_stats(_p(x), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(mean), _p(invstd),
                                    _p(a), _p(b), _p(ws), n, P, Bp, Ct, Ln, eps, momentum, _p(nbt), _stream()), "nef_bn_train_stats")
    _done(ev)
    return mean, invstd, a, b


def bn_eval_affine(gamma, beta, running_mean, running_var, eps=1e-5):
    L = _lib.load()
    Ct = gamma.numel()
    a = torch.empty(1, Ct, device=gamma.device, dtype=torch.float32)
    b = torch.empty(1, Ct, device=gamma.device, dtype=torch.float32)
    _lib.check(L.nef_bn_eval_affine(_p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(a), _p(b), Ct, eps,
                                    _stream()), "nef_bn_eval_affine")
    return a, b


def fold_bn(w, bias, a, b):
    """(a[co]*w[co], a*bias+b): eval-mode BatchNorm folded into the conv that feeds it."""
    L = _lib.load()
    _chk(w), _chk(bias)
    wf, bf = torch.empty_like(w), torch.empty_like(bias)
    _lib.check(L.nef_fold_bn(_p(w), _p(bias), _p(a), _p(b), _p(wf), _p(bf), w.shape[0], w.numel() // w.shape[0], _stream()),
               "nef_fold_bn")
    return wf, bf


def affine_relu_fwd(x, a, b, P):
    L = _lib.load()
    _chk(x)
    N, Ct, Ln = x.shape
    y = torch.empty_like(x)
    _lib.check(L.nef_affine_relu_fwd(_p(x), _p(a), _p(b), _p(y), P, N // P, Ct, Ln, _stream()), "nef_affine_relu_fwd")
    return y


def bn_relu_bwd(gy, x, gamma, mean, invstd, a, b, P, with_chan_sum=False, slots=None, phase_major=False):
    """Returns (gx, ggamma, gbeta[, sum_{b,t} gx per channel]).  `slots`: the conv_stats_buffer() the conv that produced
    `gy` filled (conv(..., bnb=...)) -- the reduction pass over (gy, x) is then skipped.  `phase_major`: gx comes back as
    [N, 2C, L/2], row 2c + p = positions p, p + 2, ... of channel c (the operand of the polyphase backward passes)."""
    L = _lib.load()
    _chk(gy), _chk(x)
    N, Ct, Ln = x.shape
    gx = torch.empty(N, 2 * Ct, Ln // 2, device=x.device, dtype=torch.float32) if phase_major else torch.empty_like(x)
    gg = torch.empty(Ct, device=x.device, dtype=torch.float32)
    gb = torch.empty(Ct, device=x.device, dtype=torch.float32)
    gs = torch.empty(Ct, device=x.device, dtype=torch.float32) if with_chan_sum else None
    n = L.nef_bn_bwd_ws_bytes(P, N // P, Ct)
    ws = workspace(n, x.device)
    ev = _hbm("bn_relu_bwd", gy, x, gx)
    fn = L.nef_bn_relu_bwd_phase_major if phase_major else L.nef_bn_relu_bwd
    _lib.check(fn(_p(gy), _p(x), _p(gamma), _p(mean), _p(invstd), _p(a), _p(b), _p(gx), _p(gg), _p(gb),
                  _p(gs), _p(ws), n, P, N // P, Ct, Ln, _p(slots[0]) if slots else None,
                  slots[1] if slots else 0, _stream()), "nef_bn_relu_bwd")
    _done(ev)
    return (gx, gg, gb, gs) if with_chan_sum else (gx, gg, gb)


def bn_relu_bwd_combine3(gy, x, mean, invstd, a, b, slots=None, phase_major=False):
    """pass_combine_bwd(bn_relu_bwd(gy, x, ..., P=3)) in one pass: returns (gP2 [2B,2C,L], ggamma, gbeta, chan sum of gx);
    `phase_major`: gP2 as [2B, 4C, L/2] (see bn_relu_bwd)."""
    L = _lib.load()
    _chk(gy), _chk(x)
    N, Ct, Ln = x.shape
    Bp = N // 3
    gP2 = torch.empty((2 * Bp, 4 * Ct, Ln // 2) if phase_major else (2 * Bp, 2 * Ct, Ln), device=x.device, dtype=torch.float32)
    gg = torch.empty(Ct, device=x.device, dtype=torch.float32)
    gb = torch.empty(Ct, device=x.device, dtype=torch.float32)
    gs = torch.empty(Ct, device=x.device, dtype=torch.float32)
    n = L.nef_bn_bwd_ws_bytes(3, Bp, Ct)
    ws = workspace(n, x.device)
    ev = _hbm("bn_relu_bwd_combine3", gy, x, gP2)
    fn = L.nef_bn_relu_bwd_combine3_phase_major if phase_major else L.nef_bn_relu_bwd_combine3
    _lib.check(fn(_p(gy), _p(x), _p(mean), _p(invstd), _p(a), _p(b), _p(gP2), _p(gg), _p(gb),
                  _p(gs), _p(ws), n, Bp, Ct, Ln, _p(slots[0]) if slots else None,
                  slots[1] if slots else 0, _stream()), "nef_bn_relu_bwd_combine3")
    _done(ev)
    return gP2, gg, gb, gs


def bn_relu_bwd_up(gu, x, mean, invstd, a, b, P, slots=None):
    """bn_relu_bwd(upsample2_bwd(gu), x, ...) with the upsampling adjoint taken on the fly; gu [N,C,2L].
    Returns (gx, ggamma, gbeta, sum_{b,t} gx per channel)."""
    L = _lib.load()
    _chk(gu), _chk(x)
    N, Ct, Ln = x.shape
    assert gu.shape == (N, Ct, 2 * Ln)
    gx = torch.empty_like(x)
    gg = torch.empty(Ct, device=x.device, dtype=torch.float32)
    gb = torch.empty(Ct, device=x.device, dtype=torch.float32)
    gs = torch.empty(Ct, device=x.device, dtype=torch.float32)
    n = L.nef_bn_bwd_ws_bytes(P, N // P, Ct)
    ws = workspace(n, x.device)
    ev = _hbm("bn_relu_bwd_up", gu, x, gx)
    _lib.check(L.nef_bn_relu_bwd_up(_p(gu), _p(x), _p(mean), _p(invstd), _p(a), _p(b), _p(gx), _p(gg), _p(gb), _p(gs),
                                    _p(ws), n, P, N // P, Ct, Ln, _p(slots[0]) if slots else None,
                                    slots[1] if slots else 0, _stream()), "nef_bn_relu_bwd_up")
    _done(ev)
    return gx, gg, gb, gs


def bn_relu_bwd_outconv(gout, out, wout, x, mean, invstd, a, b, P):
    """bn_relu_bwd(outconv_bwd_data(gout, out, wout), x, ...) without materialising the [N,C,L] gradient in between.
    Returns (gx, ggamma, gbeta, sum_{b,t} gx per channel)."""
    L = _lib.load()
    _chk(gout), _chk(out), _chk(x)
    N, Ct, Ln = x.shape
    gx = torch.empty_like(x)
    gg = torch.empty(Ct, device=x.device, dtype=torch.float32)
    gb = torch.empty(Ct, device=x.device, dtype=torch.float32)
    gs = torch.empty(Ct, device=x.device, dtype=torch.float32)
    n = L.nef_bn_bwd_outconv_ws_bytes(P, N // P, Ct, Ln)
    ws = workspace(n, x.device)
    # a two-pass pair by construction (the reduction pass reads x, the apply pass reads it again and writes gx): priced by
    # what the pair has to move -- 2 reads + 1 write of the [N, C, L] tensor (+ the small gout / out rows twice)
    ev = _hbm("bn_relu_bwd_outconv", gout, out, x, gout, out, x, gx)
    _lib.check(L.nef_bn_relu_bwd_outconv(_p(gout), _p(out), _p(wout), _p(x), _p(mean), _p(invstd), _p(a), _p(b), _p(gx),
                                         _p(gg), _p(gb), _p(gs), _p(ws), n, P, N // P, Ct, Ln, _stream()),
               "nef_bn_relu_bwd_outconv")
    _done(ev)
    return gx, gg, gb, gs


def outconv_fwd(x, w, bias, pro=None):
    """`pro` = (a, b, Bp): x is the pre-BatchNorm tensor, relu(x*a[p,c]+b[p,c]) is applied on the fly."""
    L = _lib.load()
    _chk(x), _chk(w), _chk(bias)
    N, Ct, Ln = x.shape
    out = torch.empty(N, 1, Ln, device=x.device, dtype=torch.float32)
    a, b, Bp = pro if pro is not None else (None, None, 1)
    ev = _hbm("outconv_fwd", x, out)
    _lib.check(L.nef_outconv_fwd_pro(_p(x), _p(a), _p(b), Bp, _p(w), _p(bias), _p(out), N, Ct, Ln, _stream()),
               "nef_outconv_fwd")
    _done(ev)
    return out


def outconv_bwd_data(gout, out, w, Ct):
    L = _lib.load()
    _chk(gout), _chk(out)
    N, Ln = out.shape[0], out.shape[-1]
    gx = torch.empty(N, Ct, Ln, device=out.device, dtype=torch.float32)
    _lib.check(L.nef_outconv_bwd_data(_p(gout), _p(out), _p(w), _p(gx), N, Ct, Ln, _stream()), "nef_outconv_bwd_data")
    return gx


def outconv_bwd_weight(gout, out, x, pro=None):
    L = _lib.load()
    _chk(gout), _chk(out), _chk(x)
    N, Ct, Ln = x.shape
    gw = torch.empty(1, Ct, 3, device=x.device, dtype=torch.float32)
    gb = torch.empty(1, device=x.device, dtype=torch.float32)
    n = L.nef_outconv_bwd_weight_ws_bytes(Ct)
    ws = workspace(n, x.device)
    a, b, Bp = pro if pro is not None else (None, None, 1)
    ev = _hbm("outconv_bwd_weight", gout, out, x)
    _lib.check(L.nef_outconv_bwd_weight_pro(_p(gout), _p(out), _p(x), _p(a), _p(b), Bp, _p(gw), _p(gb), _p(ws), n, N, Ct,
                                            Ln, _stream()), "nef_outconv_bwd_weight")
    _done(ev)
    return gw, gb


# ------------------------------------------------------------------ loss / optimiser
def loss_fwd(pred, pred_p, pred_l, target, factors, reg_l2, use_mask, out=None):
    L = _lib.load()
    for t in (pred, pred_p, pred_l, target):
        _chk(t)
    losses = torch.empty(4, device=pred.device, dtype=torch.float32) if out is None else out
    n = L.nef_loss_ws_bytes()
    ws = workspace(n, pred.device)
    _lib.check(L.nef_loss_fwd(_p(pred), _p(pred_p), _p(pred_l), _p(target), _p(losses), _p(ws), n, pred.numel(),
                              factors[0], factors[1], factors[2], int(reg_l2), use_mask, _stream()), "nef_loss_fwd")
    return losses


def loss_bwd(pred, pred_p, pred_l, target, gscale, factors, reg_l2, use_mask):
    L = _lib.load()
    # the three gradients as ONE [3B, 1, L] buffer (views): engine._head_bwd takes it as the stacked decoder-output gradient as is
    g3 = torch.empty((3 * pred.shape[0],) + tuple(pred.shape[1:]), device=pred.device, dtype=torch.float32)
    g_pred, g_p, g_l = g3[0:pred.shape[0]], g3[pred.shape[0]:2 * pred.shape[0]], g3[2 * pred.shape[0]:]
    _lib.check(L.nef_loss_bwd(_p(pred), _p(pred_p), _p(pred_l), _p(target), _p(gscale), _p(g_pred), _p(g_p), _p(g_l),
                              pred.numel(), factors[0], factors[1], factors[2], int(reg_l2), use_mask, _stream()),
               "nef_loss_bwd")
    return g_pred, g_p, g_l


def flatten_into(tensors, out):
    """out[:] = concatenation of `tensors` (contiguous fp32, flattened), ONE launch per 64 tensors (nef_flatten; was torch.cat)."""
    n = len(tensors)
    if n == 0:
        return out
    tensors = [t.detach() if t.is_contiguous() else t.detach().contiguous() for t in tensors]
    if not out.is_cuda or any((not t.is_cuda) or t.dtype != torch.float32 for t in tensors) or out.dtype != torch.float32:
        return torch.cat([t.reshape(-1) for t in tensors], out=out)          # (CPU plumbing tests; never on the device path)
    assert out.is_contiguous() and out.numel() == sum(t.numel() for t in tensors), (out.numel(), sum(t.numel() for t in tensors))
    srcs = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
    sizes = (C.c_int64 * n)(*[t.numel() for t in tensors])
    _lib.check(_lib.load().nef_flatten(srcs, sizes, n, _p(out), _stream()), "nef_flatten")
    return out


def stacked3(parts):
    """The three [B, 1, L] tensors as one [3B, 1, L] tensor WITHOUT a copy when they are consecutive views of one buffer (what
    loss_bwd returns), else None."""
    a, b, c = parts
    if any(t is None or not t.is_contiguous() for t in parts) or not (a.shape == b.shape == c.shape):
        return None
    n = a.numel() * 4
    if b.data_ptr() != a.data_ptr() + n or c.data_ptr() != b.data_ptr() + n or a._base is None or a._base is not b._base or a._base is not c._base:
        return None
    base = a._base
    if base.is_contiguous() and base.data_ptr() == a.data_ptr() and base.numel() == 3 * a.numel():
        return base.view((3 * a.shape[0],) + tuple(a.shape[1:]))
    return None


def regroup_halves(w, inverse=False):
    """w [Co, 2 Cih, K] -> grouped [2 Co, Cih, K] (group = input-channel half); inverse: the other way.  One launch (nef_regroup_halves)."""
    _chk(w)
    if inverse:
        co2, cih, k = w.shape
        out = torch.empty(co2 // 2, 2 * cih, k, device=w.device, dtype=torch.float32)
        co = co2 // 2
    else:
        co, ci, k = w.shape
        cih = ci // 2
        out = torch.empty(2 * co, cih, k, device=w.device, dtype=torch.float32)
    _lib.check(_lib.load().nef_regroup_halves(_p(w), _p(out), co, cih, k, int(bool(inverse)), _stream()), "nef_regroup_halves")
    return out


def view_metrics(pred, gt, rois=None):
    """Per-row PSNR and SSIM tables (fp64 [B, Q]) of pred vs gt [B, Q, L] on [0, rois[i, -1, 0])."""
    L = _lib.load()
    _chk(pred), _chk(gt)
    if rois is not None:
        _chk(rois, torch.int64)
    B, Q, Ln = pred.shape
    psnr = torch.empty(B, Q, device=pred.device, dtype=torch.float64)
    ssim = torch.empty(B, Q, device=pred.device, dtype=torch.float64)
    _lib.check(L.nef_view_metrics(_p(pred), _p(gt), _p(rois), _p(psnr), _p(ssim), B, Q, Ln, _stream()), "nef_view_metrics")
    return psnr, ssim


def sgd_momentum(p, g, buf, lr, mu, gscale, first_step, skip=None, lr_dev=None):
    """`skip`: a one-element fp32 device view (h2_taint's output, summed over the ranks): > 0 leaves p and buf untouched and counts
    the step in h2_skipped().  `lr_dev`: a one-element fp32 device tensor that replaces `lr` at run time (captured steps)."""
    L = _lib.load()
    _chk(p), _chk(buf)
    assert g.is_cuda and g.dtype == torch.float32 and g.is_contiguous()
    ev = _hbm("sgd_momentum", p, p, g, buf, buf)
    sk = _p(_amax_state(p.device)["skipped"]) if skip is not None else None
    _lib.check(L.nef_sgd_momentum(_p(p), _p(g), _p(buf), p.numel(), lr, mu, gscale, int(first_step), _p(skip), sk, _p(lr_dev),
                                  _stream()), "nef_sgd_momentum")
    _done(ev)


# ------------------------------------------------------------------ half-precision panorama decoder (pano_h.hip)
def pano_h_from_f32(x):
    """fp32 [B,C,T] -> fp16 [B,T,C] (time-major)."""
    L = _lib.load()
    _chk(x)
    B, Ct, T = x.shape
    y = torch.empty(B, T, Ct, device=x.device, dtype=torch.float16)
    _lib.check(L.nef_pano_h_from_f32(_p(x), _p(y), B, Ct, T, _stream()), "nef_pano_h_from_f32")
    return y


def pano_h_pack_weight(w):
    """fp32 [Cout,Cin,3] -> fp16 MFMA A-fragment order (flat)."""
    L = _lib.load()
    _chk(w)
    Co, Ci, K = w.shape
    assert K == 3
    wp = torch.empty(Co * Ci * 3, device=w.device, dtype=torch.float16)
    _lib.check(L.nef_pano_h_pack_weight(_p(w), _p(wp), Co, Ci, _stream()), "nef_pano_h_pack_weight")
    return wp


def pano_h_conv(x, wp, bias, Cout, N=None, upsample=False, scale=None, x_div=1, nq=1, out=None):
    """ReLU(conv_k3(pro(x)) + bias) on fp16 [.,Tin,Cin] -> fp16 [N,T,Cout].  `scale` = (tensor, sc_bs, sc_is)."""
    L = _lib.load()
    _chk(x, torch.float16), _chk(bias)
    Tin, Ci = x.shape[1], x.shape[2]
    N = x.shape[0] * x_div if N is None else N
    T = 2 * Tin if upsample else Tin
    y = torch.empty(N, T, Cout, device=x.device, dtype=torch.float16) if out is None else out
    mode = (1 if scale is not None else 0) | (2 if upsample else 0)
    sc, sc_bs, sc_is = scale if scale is not None else (None, 0, 0)
    e = _timed(("pano_h_conv", Ci, Cout, N, T))
    _lib.check(L.nef_pano_h_conv(_p(x), _p(wp), _p(bias), _p(sc), _p(y), N, T, Ci, Cout, mode, x_div, nq, sc_bs, sc_is,
                                 _stream()), "nef_pano_h_conv")
    if e is not None:
        e.record()
    return y


PANO_PAIR_MAX_T = 256     # nef_pano_h_conv_pair: up to here one 256-row tile per (sample, angle); longer sequences in tiles of 252 output rows


def pano_h_conv_pair(x, wp1, bias1, scale, wp2, bias2, N, x_div, nq, out=None):
    """Decoder layers 1 + 2 in one pass (c1 stays on chip): x fp16 [.,Tin,256] -> fp16 [N,2*Tin,128].  Up to 256 output rows one tile per
    pair; longer sequences (round 6) in tiles of 252 output rows with two recomputed slots + one halo row per side.
    `scale` = (tensor, sc_bs, sc_is) as in pano_h_conv."""
    L = _lib.load()
    _chk(x, torch.float16), _chk(bias1), _chk(bias2)
    Tin, Ci = x.shape[1], x.shape[2]
    T = 2 * Tin
    assert Ci == 256
    y = torch.empty(N, T, 128, device=x.device, dtype=torch.float16) if out is None else out
    sc, sc_bs, sc_is = scale
    e = _timed(("pano_h_conv_pair", N, T))
    _lib.check(L.nef_pano_h_conv_pair(_p(x), _p(wp1), _p(bias1), _p(sc), _p(wp2), _p(bias2), _p(y), N, T, x_div, nq,
                                      sc_bs, sc_is, _stream()), "nef_pano_h_conv_pair")
    if e is not None:
        e.record()
    return y


PANO_TAIL_MAX_T = 512     # nef_pano_h_conv_tail: up to here one tile per (sample, angle); longer sequences in tiles of 508 output rows


def pano_h_conv_tail(x, wp3, bias3, wp4, bias4, wout, bout, out, nq, out_bs, out_is):
    """Layers 3 + 4 + last conv + sigmoid(x/3) in one pass (c3, c4 stay on chip): x fp16 [N, Tin, 128] (layer 2's output) -> the fp32
    views at `out` (view base, addressed as in pano_h_conv_outconv).  Sequences of up to 512 rows are one tile per pair; longer ones run
    in tiles of 508 output rows with three recomputed halo rows per side."""
    L = _lib.load()
    _chk(x, torch.float16), _chk(bias3), _chk(bias4), _chk(wout)
    N, Tin, Ci = x.shape
    assert Ci == 128
    e = _timed(("pano_h_conv_tail", N, 2 * Tin))
    _lib.check(L.nef_pano_h_conv_tail(_p(x), _p(wp3), _p(bias3), _p(wp4), _p(bias4), _p(wout), _p(bout), _p(out), N, 2 * Tin, nq,
                                      out_bs, out_is, _stream()), "nef_pano_h_conv_tail")
    if e is not None:
        e.record()
    return out


def pano_h_outconv(x, w, bias, out, nq, out_bs, out_is):
    """out[(n/nq)*out_bs + (n%nq)*out_is + t] = sigmoid((conv_k3(x[n]) + bias)/3); x fp16 [N,T,64], out fp32 view base."""
    L = _lib.load()
    _chk(x, torch.float16), _chk(w)
    N, T, Ci = x.shape
    assert Ci == 64
    _lib.check(L.nef_pano_h_outconv(_p(x), _p(w), _p(bias), _p(out), N, T, nq, out_bs, out_is, _stream()),
               "nef_pano_h_outconv")
    return out


def pano_h_conv_outconv(x, wp, bias, wout, bout, out, nq, out_bs, out_is):
    """Layer 4 (64->64) + last conv + sigmoid(x/3) in one pass; x fp16 [N,T,64], out fp32 view base."""
    L = _lib.load()
    _chk(x, torch.float16), _chk(bias), _chk(wout)
    N, T, Ci = x.shape
    assert Ci == 64
    e = _timed(("pano_h_conv_outconv", N, T))
    _lib.check(L.nef_pano_h_conv_outconv(_p(x), _p(wp), _p(bias), _p(wout), _p(bout), _p(out), N, T, nq, out_bs, out_is,
                                         _stream()), "nef_pano_h_conv_outconv")
    if e is not None:
        e.record()
    return out
