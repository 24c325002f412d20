"""Per-step gradient check on the tests' hashed-weight model: the same model stepped with the split-fp16 convs (and whatever
NEF_* switches the environment sets, e.g. NEF_DIAG=1 NEF_POLY=0) against a clone that computes the same step on the fp32 kernels.
Shows how far a few nearly cancelling gradient sums move when activations differ in the last bit (a ReLU decision that flips):
usage: [SHAPES="[(2,512)]*4"] python tools/step_grad_check.py"""
import sys, copy, random, numpy as np, torch, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_model_gpu as tm
from electrocardio_panorama_amd import synth, ops
from electrocardio_panorama_amd.network import build_loss
from electrocardio_panorama_amd.solver.optim_scheduler import get_optimizer
DEV = "cuda"
ops._H2_MIN_WGS = 0
V = 3
cfg = tm.make_cfg(V, lr=0.05)
shapes = eval(os.environ.get("SHAPES", "[(4, 512), (2, 512), (4, 512), (2, 512)]"))
batches = [{k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in synth.make_batch(B, V, L, seed=70 + i).items()} for i, (B, L) in enumerate(shapes)]
me = tm.hashed_model(V).train(); me.dropout_p = 0.0
lossf, optim = build_loss(cfg), get_optimizer(cfg, me.parameters())
rel = lambda a, b: float((a - b).norm() / b.norm())
random.seed(3)
for i, bb in enumerate(batches):
    ref = tm.hashed_model(V).train(); ref.dropout_p = 0.0
    ref.load_state_dict(copy.deepcopy(me.state_dict()))
    st = random.getstate()
    ops.H2 = False
    o_ = ref(bb["data"], bb["input_theta"], bb["target_theta"], bb["rois"], phase="train")
    lossf(o_[0], o_[1], o_[2], bb["target_view"].unsqueeze(1), cfg)[0].backward()
    ops.H2 = True
    random.setstate(st)
    o_ = me(bb["data"], bb["input_theta"], bb["target_theta"], bb["rois"], phase="train")
    ls = lossf(o_[0], o_[1], o_[2], bb["target_view"].unsqueeze(1), cfg)
    ls[0].backward()
    gr = {k: v.grad for k, v in ref.named_parameters()}
    gm = {k: v.grad for k, v in me.named_parameters()}
    print("step", i + 1, shapes[i], {k: "%.1e" % rel(gm[k], gr[k]) for k in ("decoder.3.double_conv.4.bias", "decoder.3.double_conv.1.bias", "decoder.1.double_conv.4.bias", "decoder.1.double_conv.1.bias", "z1_conv.0.residual_conv.bias")})
    optim.step(); optim.zero_grad()
