"""Data-parallel plumbing: one process per GPU, the ECG batch sharded by rank, ONE collective per step
(sum all-reduce of the flat gradient buffer over RCCL/xGMI; backend 'nccl' is RCCL on ROCm).  Replaces the
single-process nn.DataParallel of reference codes/solver/solver.py:32-34.  BatchNorm
statistics stay per shard, as under DataParallel; rank 0's running statistics are authoritative."""
import os

import torch
from . import _env
import torch.distributed as dist


_HOOK_VARS = ("NEF_SHARE_GPU", "NEF_DIST_BACKEND", "NEF_DIST_FORCE")
_warned = []


def _hook(name):
    """Test hooks (two ranks on one GPU, gradients over gloo, a one-rank RCCL group) are honoured ONLY under
    NEF_TEST_HOOKS=1, so that a stray variable cannot put a production run on gloo; without the guard they are ignored
    with one warning."""
    v = os.environ.get(name)
    if v is None:
        return None
    if os.environ.get("NEF_TEST_HOOKS") == "1":
        return v
    if name not in _warned:
        _warned.append(name)
        import sys
        sys.stderr.write(f"[nefnet] {name}={v} ignored: test hooks need NEF_TEST_HOOKS=1\n")
    return None


def local_rank():
    """Index of the HIP device this process drives: torchrun's LOCAL_RANK (0 when ranks share one GPU under the
    NEF_SHARE_GPU test hook, or when not launched by torchrun)."""
    if _hook("NEF_SHARE_GPU") == "1":
        return 0
    return int(os.environ.get("LOCAL_RANK", "0"))


def rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env():
    """Initialise torch.distributed from torchrun's environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    Returns (rank, world, local_rank).  Single-process runs need no initialisation."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = _hook("NEF_DIST_FORCE") == "1"      # test hook: a 1-rank group still goes through RCCL
    share = _hook("NEF_SHARE_GPU") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = _hook("NEF_DIST_BACKEND")          # test hook: "gloo" lets two ranks share one GPU
        if share:
            local = 0
        if torch.cuda.is_available() and backend != "gloo":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            if torch.cuda.is_available():
                torch.cuda.set_device(local)
            dist.init_process_group("gloo", rank=rank, world_size=world)
    elif share:
        local = 0
    return rank, world, local


def shard_indices(global_batch, rank, world):
    """Contiguous equal shards; the global batch must divide evenly so that the mean of per-shard mean
    losses equals the reference's full-batch mean (solver.py:171-188)."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return list(range(rank * per, (rank + 1) * per))


def shard_batch(meta, rank, world):
    n = len(next(iter(meta.values())))
    idx = shard_indices(n, rank, world)
    return {k: v[idx[0]:idx[-1] + 1] for k, v in meta.items()}


class ShardedLoader:
    """Wraps an iterable of GLOBAL `meta` batches (identical on every rank: same seed, same order) and yields this
    rank's contiguous shard of each -- what nn.DataParallel's scatter does to the reference's batch
    (solver.py:32-34).  With world == 1 it is the identity."""

    def __init__(self, loader, rank=None, world=None):
        r, w = rank_world()
        self.loader, self.rank, self.world = loader, (r if rank is None else rank), (w if world is None else world)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for meta in self.loader:
            yield meta if self.world == 1 else shard_batch(meta, self.rank, self.world)


def reduce_flat_grads(grads, flat):
    """Pack `grads` into `flat` (one buffer) and sum it across ranks with a single all-reduce."""
    torch.cat([g.reshape(-1) for g in grads], out=flat)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat)
    return flat


# ---------------------------------------------------------------------------------------------------------------
# Early bucket: the gradients that are final long before the backward pass ends (everything except the per-lead
# encoder's: 71 % of the bytes at 3 leads) are summed across ranks WHILE the encoder blocks are still being
# back-propagated, so only the encoder bucket's all-reduce is exposed at the optimiser step.
_EARLY = {"pending": None, "backwards": 0}
TIMING = None        # bench.py: a list collecting (start_event, end_event) around the EXPOSED part of the step's all-reduce
# The early bucket makes engine.backward COLLECTIVE (every rank must run the same backward passes in the same order), so it is
# opt-in: FusedSGD -- the consumer that knows how to line the bucket up with its flat buffer -- switches it on when it is built.
# NEF_EARLY_REDUCE=0 keeps it off (one all-reduce per step).
EARLY_ENABLED = False


def enable_early_reduce():
    global EARLY_ENABLED
    EARLY_ENABLED = _env.get("NEF_EARLY_REDUCE", "1") != "0"


def early_reduce(P, grads, side_stream=None):
    """Called by engine.backward in front of the encoder blocks: packs every gradient computed so far (parameter order)
    into one buffer and starts its sum all-reduce without blocking the launching stream.  `side_stream`: the stream the
    weight gradients were issued on (the collective is ordered behind it).  No-op for a single process, under graph capture,
    and unless an optimiser opted in (enable_early_reduce).
    The bucket is a SNAPSHOT of this backward pass: it is only valid for an optimiser step that follows exactly one backward.
    A second backward before the bucket was taken (gradient accumulation, a skipped step) starts no new collective and
    poisons the pending one -- take_early() then retires it and the optimiser falls back to its single all-reduce."""
    if not EARLY_ENABLED or not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return
    if torch.cuda.is_current_stream_capturing():
        return
    _EARLY["backwards"] += 1
    if _EARLY["backwards"] > 1:
        return
    names = [n for n in P if grads.get(n) is not None]
    if not names:
        return
    if side_stream is not None:
        # some of the bucket's gradients (BatchNorm affine, theta MLP inputs) come off the launching stream: order the side
        # stream behind it before packing
        side_stream.wait_stream(torch.cuda.current_stream(side_stream.device))
    ctx = torch.cuda.stream(side_stream) if side_stream is not None else _Null()
    with ctx:
        flat = torch.cat([grads[n].reshape(-1) for n in names])
        work = dist.all_reduce(flat, async_op=True)
    _EARLY["pending"] = dict(names=names, sizes=[grads[n].numel() for n in names], flat=flat, work=work)


def take_early():
    """The pending early bucket, or None.  A bucket that more than one backward pass ran against is waited for and dropped."""
    pend, n = _EARLY["pending"], _EARLY["backwards"]
    _EARLY["pending"], _EARLY["backwards"] = None, 0
    if pend is not None and n != 1:
        pend["work"].wait()
        return None
    return pend


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class DryCollective:
    """bench.py --dry-collective N: stands in for the gradient all-reduce of an N-rank job on a ONE-GPU box (no multi-GPU node was
    available to the build).  Called like dist.all_reduce(tensor, async_op=...): instead of communicating it keeps a few workgroups
    busy on the current stream for the MODELLED duration of a ring all-reduce over xGMI (per-link bound,
    2 (N-1)/N S / 153 GB/s, + a fixed launch latency), so the overlap logic of the graphed data-parallel step -- graph A, the early
    bucket under graph B, the exposed encoder bucket -- is exercised at the real step length.  The gradients are left as they are
    (a one-rank sum)."""
    LINK_GBPS = 153.0       # one xGMI link, SURVEY.md section 5
    LATENCY_US = 20.0       # fixed cost of one collective launch (kernel launch + ring set-up), assumed

    class _Done:
        def wait(self):
            return True

    def __init__(self, ranks, wgs=8):
        self.ranks, self.wgs, self.calls = int(ranks), int(wgs), []

    def ms(self, nbytes):
        n = self.ranks
        return (2.0 * (n - 1) / n * nbytes / (self.LINK_GBPS * 1e9)) * 1e3 + self.LATENCY_US * 1e-3

    def __call__(self, t, async_op=False):
        from . import _lib
        ms = self.ms(t.numel() * t.element_size())
        self.calls.append((t.numel() * t.element_size(), ms))
        L = _lib.load()
        _lib.check(L.nef_debug_spin_us(float(ms * 1e3), self.wgs, torch.cuda.current_stream().cuda_stream), "nef_debug_spin_us")
        return self._Done() if async_op else None


def sum_counts(counts, device):
    """Sum of a short list of host integers over the ranks (e.g. the split-fp16 clamp / skipped-step counters): every rank
    gets the same totals, so a decision taken on them -- raise, warn, carry on -- is taken by ALL ranks together and none is left
    blocking in the next collective.  World size 1: the list itself."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return [int(c) for c in counts]
    on_dev = dist.get_backend() == "nccl"
    t = torch.tensor([int(c) for c in counts], dtype=torch.int64, device=device if on_dev else "cpu")
    dist.all_reduce(t)
    return [int(v) for v in t.tolist()]


def broadcast_buffers(model, src=0):
    """Make rank `src`'s BatchNorm running statistics authoritative (e.g. before a checkpoint)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for b in model.buffers():
            dist.broadcast(b, src)
