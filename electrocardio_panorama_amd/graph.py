"""hipGraph-captured train step.

At the reference's own training shape (batch 32, 512-sample beats, `train_net.py:27`) a step is ~250 short launches
and the host cannot issue them as fast as the GPU retires them.  `GraphedTrainStep` captures forward + losswrapper +
backward + momentum-SGD once per input shape (torch.cuda.CUDAGraph on the stream the C ABI launches into) and replays
it: inputs are copied into static buffers, and the only per-step host decisions -- the two Standin lead choices
(reference model_nefnet.py:154,156, drawn from Python's `random` in the reference's order) and the dropout seed --
travel through device words that the kernels read at run time.  Same arithmetic as the eager path.

One graph (with its own static input buffers) is kept per input shape, so a final partial batch or alternating shapes
replay instead of re-capturing; the flat parameter / gradient / momentum buffers are shared by all of them and survive
every re-capture (shape change); the learning rate lives in a device word the captured SGD launch reads, so a scheduler step costs
one small copy and no re-capture.  `state_dict()` / `load_state_dict()` carry the momentum buffer
for checkpoints.

Data parallel (world > 1): the step is captured as TWO graphs cut at the early-bucket point of the backward pass; the all-reduce
of the bucket that is final there runs between the replays, under the second graph (`_capture_split`; NEF_GRAPH_SPLIT=0 keeps one
graph and one exposed all-reduce).
"""
import random

import torch
from . import _env
import torch.distributed as dist

from . import engine, ops


class GraphedTrainStep:
    def __init__(self, model, cfg, lr=None, momentum=0.9, optimizer=None):
        """`optimizer`: a solver.optim_scheduler.FusedSGD over model.parameters() -- the graph then steps THAT optimiser's flat
        parameter / momentum buffers (so checkpoints, a learning-rate scheduler and eager steps in between see one state) and
        takes lr / momentum from its parameter group; without it the stepper owns its buffers (lr / momentum arguments)."""
        if cfg.DATA.noise:
            raise NotImplementedError("cfg.DATA.noise adds a host-side tensor op between model and loss")
        self.model, self.cfg = model, cfg
        self.optimizer = optimizer
        if optimizer is not None:
            if len(optimizer.param_groups) != 1 or not hasattr(optimizer, "_flat"):
                raise NotImplementedError("the graphed step drives FusedSGD with one parameter group")
            lr, momentum = optimizer.param_groups[0]["lr"], optimizer.param_groups[0]["momentum"]
        self.lr = float(cfg.SOLVER.lr if lr is None else lr)
        self.mu = float(momentum)
        self.factors = tuple(float(f) for f in cfg.SOLVER.loss_factor)
        self.reg_l2 = {"l1_loss": False, "l2_loss": True}[cfg.SOLVER.reg_loss]
        u = cfg.SOLVER.loss_using
        self.use_mask = (1 if 1 in u else 0) | (2 if 2 in u else 0) | (4 if 3 in u else 0)
        self.slots = {}          # input shape -> static input buffers + captured graph
        self.flat_p = self.flat_g = self.flat_buf = None
        self.live = None
        self.choice_dev = None
        self.lr_dev = None
        self.calls = 0
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # dp: the step sums its gradients through torch.distributed.  True for more than one rank -- and for a ONE-rank group under the
        # NEF_DIST_FORCE test hook, so that the collectives between the two graph replays run on the real backend (RCCL) of a one-GPU box
        from . import parallel as _par
        self.dp = self.world > 1 or (dist.is_available() and dist.is_initialized() and _par._hook("NEF_DIST_FORCE") == "1")
        # data parallel: capture the step as two graphs with the early gradient bucket's all-reduce between them (_capture_split)
        self.split_capture = _env.get("NEF_GRAPH_SPLIT", "1") != "0"
        # bench.py --dry-collective: a parallel.DryCollective standing in for dist.all_reduce on a one-GPU box (set `dp` with it)
        self.dry = None

    def _all_reduce(self, t, async_op=False):
        if self.dry is not None:
            return self.dry(t, async_op)
        return dist.all_reduce(t, async_op=async_op)

    # -------------------------------------------------------------------------------------------------
    def _flatten(self, live):
        """Parameters become views of one flat buffer.  Done once: the live-parameter set does not depend on the input
        shape, so later captures reuse the buffers -- and with them the momentum."""
        named = dict(self.model.named_parameters())
        if self.optimizer is not None:
            # the optimiser's own flat buffers (FusedSGD._build: parameters become views of fl["p"], the momentum views of
            # fl["buf"] live in optimizer.state): nothing to copy, nothing to keep in sync
            opt, params = self.optimizer, [named[k] for k in live]
            fl = opt._flat.get(0)
            if fl is None or fl["ids"] != [id(p) for p in params] or any(
                    p.data.data_ptr() < fl["p"].data_ptr() or
                    p.data.data_ptr() >= fl["p"].data_ptr() + fl["p"].numel() * 4 for p in params):
                opt._build(0, params)
                fl = opt._flat[0]
            self.live = live
            self.flat_p, self.flat_g, self.flat_buf, self.flat_g_all = fl["p"], fl["g"], fl["buf"], fl["g_all"]
            return
        if self.live == live and self.flat_p is not None and all(
                named[k].data.data_ptr() >= self.flat_p.data_ptr() and
                named[k].data.data_ptr() < self.flat_p.data_ptr() + 4 * self.flat_p.numel() for k in live):
            return
        old_buf, old_live = self.flat_buf, self.live
        self.live = live
        n = sum(named[k].numel() for k in live)
        dev = self.data.device
        self.flat_p = torch.empty(n, device=dev, dtype=torch.float32)
        from .solver.optim_scheduler import GRAD_HDR
        self.flat_g_all = torch.zeros(n + GRAD_HDR, device=dev, dtype=torch.float32)      # [header: taint word + 3 zeros | gradients] (FusedSGD._build)
        self.flat_g = self.flat_g_all[GRAD_HDR:]
        self.flat_buf = torch.zeros(n, device=dev, dtype=torch.float32)
        if old_buf is not None and old_live == live:       # parameters were re-pointed from outside: keep the momentum
            self.flat_buf.copy_(old_buf)
        off = 0
        for k in live:
            p = named[k]
            m = p.numel()
            self.flat_p[off:off + m].copy_(p.data.reshape(-1))
            old_ptr = p.data.data_ptr()
            p.data = self.flat_p[off:off + m].view_as(p.data)       # parameters become views of the flat buffer
            ops.amax_move(old_ptr, p.data.data_ptr())
            off += m

    def _fwd_bwd(self):
        m = self.model
        P = {k: v.detach() for k, v in m.named_parameters()}
        drop = engine.DropCfg(True, m.dropout_p, m.dropout_masks, seed=0, seed_dev=self.seed_dev)
        with ops.amax_scope((m._nef_scope, True)):        # the train-mode call sites of the model (Model_nefnet._engine_fwd)
            outs, sv = engine.forward(P, dict(m.named_buffers()), self.data, self.in_theta, self.q_theta, self.rois,
                                      phase="train", training=True, drop=drop, lead_choice=self.choice_dev, save=True,
                                      status=self.status)
            o, p_, l_ = (t.contiguous() for t in outs)
            ops.loss_fwd(o, p_, l_, self.target, self.factors, self.reg_l2, self.use_mask, out=self.losses)
            g3 = ops.loss_bwd(o, p_, l_, self.target, None, self.factors, self.reg_l2, self.use_mask)
            return engine.backward(P, sv, g3)

    def _sgd(self):
        # the learning rate travels through a device word (a captured launch freezes its scalars): a scheduler step updates the word,
        # nothing is re-captured
        ops.sgd_momentum(self.flat_p, self.flat_g, self.flat_buf, self.lr, self.mu, 1.0 / self.world, False,
                         skip=self.flat_g_all[:1], lr_dev=self.lr_dev)

    def _body(self):
        grads = self._fwd_bwd()
        ops.flatten_into([grads[k] for k in self.live], self.flat_g)
        ops.h2_taint(self.flat_g_all[:1])      # this step's clamped split-fp16 launches: the update is skipped (on every rank)
        if not self.dp:
            self._sgd()

    def _capture_split(self):
        """Data parallel: the step as TWO graphs, cut where engine.backward would start the early gradient bucket (everything
        behind the per-lead encoder: a suffix of the parameter order, 71 % of the bytes at 3 leads).  Between the two replays the
        suffix bucket's all-reduce is started on a side stream and runs under the second graph -- the encoder blocks' backward
        pass, ~10 ms at configs[1] -- so that only the encoder bucket (+ the taint word in front of it) is summed behind the
        replay: the overlap the eager path has had since round 3 (parallel.early_reduce), bit for bit the same buffer."""
        import gc
        gA, gB = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        info = {}

        def hook(P, grads, side):
            side.join()                                      # nothing forked may be open when a capture ends
            early = [k for k in self.live if grads.get(k) is not None]
            k0 = len(self.live) - len(early)
            if self.live[k0:] != early:
                raise RuntimeError("early gradient bucket is not a suffix of the live parameters")
            named = dict(self.model.named_parameters())
            info["k0"], info["split"] = k0, sum(named[k].numel() for k in self.live[:k0])
            ops.flatten_into([grads[k] for k in early], self.flat_g[info["split"]:])
            gA.capture_end()
            gB.capture_begin(pool=gA.pool())

        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.empty_cache()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            gA.capture_begin()
            engine.EARLY_HOOK = hook
            try:
                grads = self._fwd_bwd()
            finally:
                engine.EARLY_HOOK = None
            if "split" not in info:
                raise RuntimeError("engine.backward never reached its early-bucket point")
            if info["k0"]:
                ops.flatten_into([grads[k] for k in self.live[:info["k0"]]], self.flat_g[:info["split"]])
            ops.h2_taint(self.flat_g_all[:1])
            gB.capture_end()
        torch.cuda.current_stream().wait_stream(cap)
        self._comm = getattr(self, "_comm", None) or torch.cuda.Stream()
        return (gA, gB, info["split"])

    def _use(self, slot):
        self.data, self.in_theta, self.q_theta, self.rois, self.target = (slot[k] for k in
                                                                          ("data", "in_theta", "q_theta", "rois", "target"))

    def _build(self, data, in_theta, q_theta, rois, target):
        dev = data.device
        # static input buffers: contiguous whatever the caller's strides are (a sharded loader hands out slices)
        slot = dict(data=torch.empty(data.shape, device=dev, dtype=torch.float32),
                    in_theta=torch.empty(in_theta.shape, device=dev, dtype=torch.float32),
                    q_theta=torch.empty(q_theta.shape, device=dev, dtype=torch.float32),
                    rois=torch.empty(rois.shape, device=dev, dtype=torch.int64),
                    target=torch.empty(data.shape[0], 1, data.shape[2], device=dev, dtype=torch.float32))
        self._use(slot)
        if self.choice_dev is None:
            self.choice_dev = torch.zeros(2, device=dev, dtype=torch.int32)
            self.seed_dev = torch.zeros(1, device=dev, dtype=torch.int64)
            self.status = torch.zeros(1, device=dev, dtype=torch.int32)
            self.losses = torch.zeros(4, device=dev, dtype=torch.float32)
            self.lr_dev = torch.full((1,), self.lr, device=dev, dtype=torch.float32)
        # the probe runs THIS step's inputs, Standin choices and dropout seed (drawn by __call__ before it builds): the split-fp16
        # convs measure their operands in it, and what they measure must be what the eager path measures on the same step
        self._stage(data, in_theta, q_theta, rois, target, draw=False)
        self.model.train()
        # eager probe (no update): which parameters are live, and every kernel variant gets its one-time setup
        saved = {k: v.clone() for k, v in self.model.named_buffers()}
        # ... and no history either: the probe rolls and re-measures the operand magnitudes of the split-fp16 call sites; the sites
        # that existed before it get back what they held, so the step that follows rolls them exactly once -- as the eager path and
        # an already captured shape do.  (Without this a stepper restored from a checkpoint rolled twice on its first step and was
        # not bit-identical to the uninterrupted run; a fresh run is unaffected: rolling the same pair twice is rolling it once.)
        amax0 = ops.amax_snapshot(dev)
        grads = self._fwd_bwd()
        ops.amax_restore(dev, amax0)
        if self.dp:                          # the eager probe may have started an early gradient bucket: retire it
            from . import parallel
            pend = parallel.take_early()
            if pend is not None:
                pend["work"].wait()
        self._flatten([k for k, _ in self.model.named_parameters() if grads.get(k) is not None])
        for k, v in self.model.named_buffers():
            v.copy_(saved[k])
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        saved = {k: v.clone() for k, v in self.model.named_buffers()}
        p0, b0 = self.flat_p.clone(), self.flat_buf.clone()
        if self.dp and self.split_capture:
            graph = self._capture_split()
        else:
            with torch.cuda.graph(graph):
                self._body()
        # capture does not execute, but keep state exactly as before the capture regardless
        for k, v in self.model.named_buffers():
            v.copy_(saved[k])
        self.flat_p.copy_(p0)
        self.flat_buf.copy_(b0)
        slot["graph"] = graph
        # The graph bakes in the pointers of every scratch buffer its launches used (ops.workspace, incl. the side
        # stream's, which the eager probe allocated from the general pool).  ops.workspace() REPLACES a buffer when a later
        # probe of a larger shape outgrows it; holding the buffers here keeps the replaced ones alive for as long as this
        # graph can be replayed.
        slot["workspaces"] = list(ops._WS.values())
        return slot

    def _draw(self):
        """This step's host decisions: Python `random` consumed exactly twice, z1 choice first (model_nefnet.py:154,156)."""
        V = self.model.lead_num
        self.calls += 1
        self._draws = (random.randint(0, V - 1), random.randint(0, V - 1))

    def _stage(self, data, in_theta, q_theta, rois, target, draw=True):
        self.data.copy_(data, non_blocking=True)
        self.in_theta.copy_(in_theta, non_blocking=True)
        self.q_theta.copy_(q_theta, non_blocking=True)
        self.rois.copy_(rois, non_blocking=True)
        self.target.copy_(target.reshape(self.target.shape), non_blocking=True)
        if draw:
            self._draw()
        rank = dist.get_rank() if self.world > 1 else 0
        # same ingredients as the eager path (model_nefnet.py: initial seed + call counter + epoch + rank): a resumed run
        # (Solver sets model.dropout_epoch; `calls` travels in state_dict) does not replay the masks of step 1
        seed = (torch.initial_seed() + self.calls + int(getattr(self.model, "dropout_epoch", 0)) * 0x1000003
                + rank * 0x9E3779B1) & 0x7FFFFFFFFFFF
        c1, c2 = getattr(self, "_draws", (0, 0))
        # one asynchronous launch carrying the three values as kernel arguments (rounds 1-5: two BLOCKING host-to-device copies,
        # which made the host wait for the previous step's replay before it could stage this one)
        from . import _lib
        _lib.check(_lib.load().nef_step_words(self.choice_dev.data_ptr(), self.seed_dev.data_ptr(), int(c1), int(c2), int(seed),
                                              torch.cuda.current_stream().cuda_stream), "nef_step_words")

    # -------------------------------------------------------------------------------------------------
    def set_lr(self, lr):
        """The captured SGD launch reads its learning rate from a device word: a new value is one small copy, no re-capture."""
        if float(lr) != self.lr:
            self.lr = float(lr)
            if getattr(self, "lr_dev", None) is not None:
                self.lr_dev.fill_(self.lr)

    def state_dict(self):
        """The optimiser state of the graphed path: the flat momentum buffer and the parameter order it refers to."""
        return {"lr": self.lr, "momentum": self.mu, "live": list(self.live or []), "calls": int(self.calls),
                "momentum_buffer": None if self.flat_buf is None else self.flat_buf.detach().cpu().clone()}

    def load_state_dict(self, sd):
        self.set_lr(sd["lr"])
        self.mu = float(sd["momentum"])
        self.calls = int(sd.get("calls", 0))        # the dropout seed's step counter
        self._pending_momentum = (list(sd["live"]), sd["momentum_buffer"])
        if self.flat_buf is not None:
            self._restore_momentum()

    def _restore_momentum(self):
        pend = getattr(self, "_pending_momentum", None)
        if pend is None or pend[1] is None:
            return
        live, buf = pend
        if live != self.live or buf.numel() != self.flat_buf.numel():
            raise ValueError("momentum buffer does not match this model's live parameters")
        self.flat_buf.copy_(buf.to(self.flat_buf.device))
        self._pending_momentum = None

    def __call__(self, data, in_theta, q_theta, rois, target):
        """One train step; returns the device tensor [loss, f0*l1, f1*l2, f2*l3] (valid in stream order)."""
        self.model._check_inputs(data, rois)       # what Model_nefnet.forward rejects (float rois, L % 4, CPU tensors) is rejected here too
        if self.optimizer is not None:
            g = self.optimizer.param_groups[0]
            if float(g["momentum"]) != self.mu:                # (momentum is a captured scalar: re-capture)
                self.mu = float(g["momentum"])
                self.slots.clear()
            self.set_lr(g["lr"])                               # a scheduler stepped: the device word follows, nothing is re-captured
            fl = self.optimizer._flat.get(0)
            if self.flat_p is not None and (fl is None or fl["p"] is not self.flat_p):   # e.g. optimizer.load_state_dict
                self.slots.clear()
        gen = ops.amax_generation(data.device)
        if getattr(self, "_amax_gen", gen) != gen:        # the split-fp16 site table started over: captured launches hold stale slots
            self.slots.clear()
        self._amax_gen = gen
        if getattr(self, "_scope", None) != self.model._nef_scope:       # the model loaded other weights: its split-fp16 call sites
            self._scope = self.model._nef_scope                            # start over (ops.amax_scope), and the captures hold the old slots
            self.slots.clear()
        self._draw()
        shape = (tuple(data.shape), tuple(in_theta.shape))
        slot = self.slots.get(shape)
        if slot is None:
            slot = self.slots[shape] = self._build(data, in_theta, q_theta, rois, target)
            self._restore_momentum()
        self._use(slot)
        self._stage(data, in_theta, q_theta, rois, target, draw=False)
        if not self.dp:
            slot["graph"].replay()
            return self.losses
        from . import parallel
        timing = parallel.TIMING is not None       # bench.py: HIP events around what the launching stream waits for
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if timing else None
        if isinstance(slot["graph"], tuple):
            gA, gB, split = slot["graph"]
            gA.replay()
            cur = torch.cuda.current_stream()
            self._comm.wait_stream(cur)
            with torch.cuda.stream(self._comm):        # the suffix bucket travels while graph B (the encoder's backward pass) runs
                work = self._all_reduce(self.flat_g[split:], async_op=True)
            gB.replay()
            if ev is not None:
                ev[0].record()
            self._all_reduce(self.flat_g_all[:self.flat_g_all.numel() - self.flat_g.numel() + split])       # the encoder bucket, the header (taint word) in front of it
            work.wait()
            cur.wait_stream(self._comm)
        else:
            slot["graph"].replay()
            if ev is not None:
                ev[0].record()
            self._all_reduce(self.flat_g_all)                  # one fully exposed all-reduce (NEF_GRAPH_SPLIT=0)
        if ev is not None:
            ev[1].record()
            parallel.TIMING.append(ev)
        self._sgd()
        return self.losses
