"""`Model_nefnet` with the reference's module API on top of the HIP engine.

Drop-in for reference codes/network/model_nefnet.py:63-218: same constructor arguments, same
`forward(x, input_thetas, query_theta, rois, rest_theta=None, phase='train')` / `gen_ecg(...)`
signatures, phases and return tuples, same `state_dict()` keys (so reference checkpoints load both
ways).  The torch.nn layers below are parameter containers only -- they give the reference's names,
shapes and default initialisation; their own forward() is never called.  All compute goes through
`engine` -> `ops` -> libnefnet_hip.so, and raises if that library is missing.
"""
import math
import random

import torch
import torch.nn as nn

from .. import engine, ops


class _Block(nn.Module):
    """Parameter container of BasicBlock (model_nefnet.py:36-47 / encoder/resnet_1d.py:27-37)."""

    def __init__(self, cin, cout, groups, k, residual_conv):
        super().__init__()
        self.conv1 = nn.Conv1d(cin, cout, k, 1, k // 2, bias=False, groups=groups)
        self.conv2 = nn.Conv1d(cout, cout, k, 1, k // 2, bias=False, groups=groups)
        if residual_conv:
            self.residual_conv = nn.Conv1d(cin, cout, 1, 1, groups=groups)


class _Encoder(nn.Module):
    """Live part of Encoder(backbone='resnet34') (encoder/encoder.py:19-24): conv1 + layer1 (3 blocks)."""

    def __init__(self, lead_num, init_channels=128):
        super().__init__()
        c = init_channels * lead_num
        self.conv1 = nn.Conv1d(lead_num, c, 15, 2, 7, bias=False, groups=lead_num)
        self.layer1 = nn.Sequential(*[_Block(c, c, lead_num, 7, False) for _ in range(3)])
        for m in self.modules():                                  # resnet_1d.py:114-117
            if isinstance(m, nn.Conv1d):
                n = m.kernel_size[0] * m.kernel_size[0] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))


class _DoubleConv(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.double_conv = nn.Sequential(nn.Conv1d(cin, cout, 3, padding=1), nn.BatchNorm1d(cout), nn.Identity(),
                                         nn.Conv1d(cout, cout, 3, padding=1), nn.BatchNorm1d(cout), nn.Identity())


class _NefNetFn(torch.autograd.Function):
    """One autograd node for the whole train-phase forward; backward is engine.backward."""

    @staticmethod
    def forward(ctx, model, x, in_theta, q_theta, rois, choice, drop, names, *params):
        P = dict(zip(names, params))
        outs, sv = model._engine_fwd(P, model._buffers_by_name(), x, in_theta, q_theta, rois, phase="train",
                                     training=model.training, drop=drop, lead_choice=choice, save=True,
                                     status=model._status)
        ctx.sv, ctx.names, ctx.P, ctx.bwd = sv, names, P, model._engine_bwd
        if model.keep_saved:
            model.last_saved = sv
        return outs

    @staticmethod
    def backward(ctx, g_out, g_p, g_l):
        grads = ctx.bwd(ctx.P, ctx.sv, (g_out, g_p, g_l))
        ctx.sv = None
        return (None,) * 8 + tuple(grads.get(n) for n in ctx.names)


class Model_nefnet(nn.Module):
    """Nef-Net (reference codes/network/model_nefnet.py:63)."""
    _ENGINE = (engine.forward, engine.backward)

    def _engine_fwd(self, *args, **kw):
        # the split-fp16 convs keep their operand magnitudes per model AND per mode (ops.py): train-mode and eval-mode passes see
        # different activations (batch vs running BatchNorm statistics, dropout), and a scale that follows every 2 x change
        # (ops.H2_FOLLOW_UP) would otherwise flip between them -- a repeated train step would not find the scales it left
        with ops.amax_scope((self._nef_scope, bool(kw.get("training", False)))):
            return self._ENGINE[0](*args, **kw)

    def _engine_bwd(self, *args, **kw):
        with ops.amax_scope((self._nef_scope, True)):
            return self._ENGINE[1](*args, **kw)

    def load_state_dict(self, *args, **kw):
        self._nef_scope = ops.new_amax_scope()      # other weights, other magnitudes: measure again (or load_h2_state)
        return super().load_state_dict(*args, **kw)

    def h2_state(self):
        """The operand magnitudes of this model's split-fp16 call sites (ops.h2_export) -- not part of the reference's state_dict
        (codes/utils/checkpointer.py:24-36 saves model / optimizer / scheduler); CheckPointer stores it next to them so that a
        resumed run continues bit for bit."""
        return ops.h2_export(self._nef_scope, {p.data_ptr(): n for n, p in self.named_parameters()})

    def load_h2_state(self, blob):
        """After load_state_dict: the call sites take up the magnitudes the checkpointed run had reached."""
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            return 0
        return ops.h2_import(self._nef_scope, {n: p.data_ptr() for n, p in self.named_parameters()}, blob, dev)

    def __init__(self, theta_encoder_len=1, lead_num=1):
        super().__init__()
        self._nef_scope = ops.new_amax_scope()
        if theta_encoder_len != 1:
            # ThetaEncoder.forward ignores encoder_len (theta_encoder.py:13-29), so mlp1/mlp2 only fit theta_L == 1
            raise ValueError("theta_encoder_len must be 1 (the reference's angular encoding emits 12 values)")
        self.lead_num = lead_num
        V = self._group_leads(lead_num)          # leads the grouped encoder layers are built for
        self.W_encoder = _Encoder(V, 128)
        self.mlp1 = nn.Linear(12, 128)
        self.mlp2 = nn.Linear(12, 256)
        self.w_feature_extractor = nn.Sequential(nn.Conv1d(128, 128, 3, 1, 1), nn.Identity())   # never used (:79)
        self.w_conv = nn.Sequential(_Block(128 * V, 128 * V, V, 3, True))
        self.z1_conv = nn.Sequential(_Block(64 * V, 128 * V, V, 3, True))
        self.z2_conv1 = nn.Sequential(_Block(64 * V, 128 * V, V, 3, True))
        self.z2_conv2 = nn.Sequential(
            _Block(896 * V, 896 * V, 7 * V, 3, True),
            nn.ConvTranspose1d(896 * V, 448 * V, kernel_size=2, stride=2, groups=7 * V),
            _Block(448 * V, 896 * V, 7 * V, 3, True))
        self.decoder = nn.Sequential(nn.Identity(), _DoubleConv(256, 128), nn.Identity(), _DoubleConv(128, 64),
                                     nn.Conv1d(64, 1, 3, padding=1))
        self._extra_layers()
        for n, p in self.named_parameters():
            p._nef_name = n                # lets FusedSGD line its flat buffer up with the early gradient bucket (parallel.py)
        self.dropout_p = engine.DROP_P
        self.dropout_masks = None      # test hook: {site: uint8 keep-mask} replayed instead of the RNG
        self.keep_saved = False        # test hook: expose the saved forward state of the last train-phase call
        self.last_saved = None
        # 'fp32' (reference arithmetic) or 'fp16': eval-mode view sweeps (phase 'val'/'test' rest_out, gen_ecg) on the
        # fp16 matrix cores with fp32 accumulation -- opt-in, gated at 2e-3 rel-L2 against the fp32 path
        self.panorama_dtype = 'fp32'
        self._drop_calls = 0
        self.dropout_epoch = 0         # set by Solver.train: a resumed run must not replay the masks of epoch 0
        self._status = None

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _group_leads(lead_num):
        return lead_num

    def _extra_layers(self):
        pass

    def _params_by_name(self):
        return dict(self.named_parameters())

    def _buffers_by_name(self):
        return dict(self.named_buffers())

    def _check_inputs(self, x, rois):
        if not x.is_cuda:
            raise RuntimeError("Model_nefnet runs on a HIP device only (no CPU path); move model and inputs to cuda")
        if rois.dtype.is_floating_point:
            raise TypeError("rois must be an integer tensor [B,7,2] (the reference mutates float rois in place)")
        if x.shape[-1] % 4 != 0:
            raise ValueError("signal length must be a multiple of 4")

    @staticmethod
    def _f32(t):
        return t.detach().to(torch.float32).contiguous()

    def _drop_cfg(self):
        """Counter-RNG seed of this forward: global seed + epoch + call counter, offset per data-parallel rank so that
        shards draw independent masks (parameters and Standin lead choices stay identical across ranks)."""
        self._drop_calls += 1
        rank = torch.distributed.get_rank() if (torch.distributed.is_available() and
                                                torch.distributed.is_initialized()) else 0
        return engine.DropCfg(self.training, self.dropout_p, self.dropout_masks,
                              seed=(torch.initial_seed() + self._drop_calls + int(self.dropout_epoch) * 0x1000003
                                    + rank * 0x9E3779B1) & 0x7FFFFFFFFFFF)

    def _half_sweep(self):
        if self.panorama_dtype not in ('fp32', 'fp16'):
            raise ValueError(f"panorama_dtype must be 'fp32' or 'fp16', got {self.panorama_dtype!r}")
        return self.panorama_dtype == 'fp16'

    def segment_status(self):
        """1 if any forward saw ROIs whose latent segment lengths were negative or did not sum to T (device flag,
        read lazily: this call synchronises)."""
        return 0 if self._status is None else int(self._status.item())

    # ------------------------------------------------------------------ reference API
    def forward(self, x, input_thetas, query_theta, rois, rest_theta=None, phase='train'):
        self._check_inputs(x, rois)
        if self._status is None or self._status.device != x.device:
            self._status = torch.zeros(1, dtype=torch.int32, device=x.device)
        V = self.lead_num
        x, input_thetas, query_theta = self._f32(x), self._f32(input_thetas), self._f32(query_theta)
        rois = rois.detach().to(torch.int64).contiguous()
        drop = self._drop_cfg()
        if phase == 'gen':
            with torch.no_grad():
                (z1, z2), _ = self._engine_fwd(self._params_by_name(), self._buffers_by_name(), x, input_thetas,
                                             query_theta, rois, phase='gen', training=self.training, drop=drop)
            return z1, z2
        # Python `random` is consumed exactly twice, z1 choice first (model_nefnet.py:154,156)
        choice = (random.randint(0, V - 1), random.randint(0, V - 1))
        if phase == 'train':
            named = [(n, p) for n, p in self.named_parameters()]
            names = tuple(n for n, _ in named)
            if torch.is_grad_enabled() and any(p.requires_grad for _, p in named):
                return _NefNetFn.apply(self, x, input_thetas, query_theta, rois, choice, drop, names,
                                       *[p for _, p in named])
            with torch.no_grad():
                outs, _ = self._engine_fwd(dict(named), self._buffers_by_name(), x, input_thetas, query_theta, rois,
                                         phase='train', training=self.training, drop=drop, lead_choice=choice,
                                         status=self._status)
            return outs
        if phase in ('val', 'test'):
            with torch.no_grad():
                outs, _ = self._engine_fwd(self._params_by_name(), self._buffers_by_name(), x, input_thetas, query_theta,
                                         rois, rest_theta=self._f32(rest_theta), phase=phase, training=self.training,
                                         drop=drop, lead_choice=choice, status=self._status,
                                         half_sweep=self._half_sweep())
            return outs
        raise KeyError("please type correct phase")

    def gen_ecg(self, z1, z2, query_theta, rois):
        self.eval()
        with torch.no_grad(), ops.amax_scope((self._nef_scope, False)):      # the sweep's split-fp16 convs keep their call sites (no measuring launch per call)
            return engine.gen_ecg(self._params_by_name(), self._buffers_by_name(), self._f32(z1), self._f32(z2),
                                  self._f32(query_theta), rois.detach().to(torch.int64).contiguous(),
                                  half=self._half_sweep())


class Model_nefnet2(Model_nefnet):
    """Nef-Net2 (reference codes/network/model_nefnet2.py:63-228): one single-lead encoder shared by every input lead,
    plus `single_conv_z1` / `single_conv_z2`.  Same forward / gen_ecg signatures and return tuples as Model_nefnet,
    except that phase 'gen' returns the two lead MEANS `(z1_mean, z2_mean)`, each [B, 128, T] (model_nefnet2.py:158-159).
    Not reachable from the reference's config (`build_model` only knows 'model_nefnet'); offered under the name
    'model_nefnet2'.  `dropout_masks`, when set, are per folded sample (lead-major, V*B rows)."""
    _ENGINE = (engine.forward2, engine.backward2)

    @staticmethod
    def _group_leads(lead_num):
        return 1

    def _extra_layers(self):
        self.single_conv_z1 = nn.Sequential(nn.Conv1d(128, 128, 3, 1, 1))
        self.single_conv_z2 = nn.Sequential(nn.Conv1d(128, 128, 3, 1, 1))
