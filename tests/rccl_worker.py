"""Worker of test_rccl_single_rank: a 1-rank process group on the REAL backend ('nccl' = RCCL) running the collectives
the data-parallel step uses (flat-gradient all-reduce inside FusedSGD.step, barrier, buffer broadcast)."""
import os
import random
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from electrocardio_panorama_amd import parallel, synth                       # noqa: E402
from electrocardio_panorama_amd.network import build_loss, build_model       # noqa: E402
from electrocardio_panorama_amd.solver.optim_scheduler import get_optimizer  # noqa: E402
from test_model_gpu import make_cfg                                          # noqa: E402

os.environ["NEF_DIST_FORCE"] = "1"
os.environ["NEF_TEST_HOOKS"] = "1"
rank, world, local = parallel.init_from_env()
assert dist.is_initialized() and dist.get_backend() == "nccl", dist.get_backend()
dev = torch.device("cuda", local)
cfg = make_cfg(3)
torch.manual_seed(1)
random.seed(1)
model = build_model(cfg).float().to(dev).train()
model.dropout_p = 0.0
lossf, optim = build_loss(cfg), get_optimizer(cfg, model.parameters())
b = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth.make_batch(4, 3, 512, seed=2).items()}
before = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
out = model(b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="train")
lossf(*out, b["target_view"].unsqueeze(1), cfg)[0].backward()
live = [p for p in model.parameters() if p.grad is not None]
flat = torch.empty(sum(p.numel() for p in live), device=dev)
expect = torch.cat([p.grad.reshape(-1) for p in live]).clone()
parallel.reduce_flat_grads([p.grad for p in live], flat)                     # packs; all-reduce only when world > 1
dist.all_reduce(flat)                                                        # ... so issue the RCCL collective itself
dist.barrier()
assert torch.equal(flat, expect)                                             # sum over one rank
optim.step()
parallel.broadcast_buffers(model)
for bf in model.buffers():
    dist.broadcast(bf, 0)
torch.cuda.synchronize()
after = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
assert not torch.equal(before, after) and bool(torch.isfinite(after).all())
# the GRAPHED data-parallel step on the real backend: two captured graphs, the early gradient bucket's RCCL all-reduce started on the
# communication stream between the replays, the encoder bucket (+ taint word) reduced behind the second replay -- against eager
# FusedSGD steps on a copy of the model (bit-identical parameters, as in the gloo world-2 test)
import copy                                                                   # noqa: E402
from electrocardio_panorama_amd.graph import GraphedTrainStep                 # noqa: E402
m_e = build_model(cfg).float().to(dev).train()
m_e.load_state_dict(copy.deepcopy(model.state_dict()))
m_g = build_model(cfg).float().to(dev).train()
m_g.load_state_dict(copy.deepcopy(model.state_dict()))
m_e.dropout_p = m_g.dropout_p = 0.0
o_e, o_g = get_optimizer(cfg, m_e.parameters()), get_optimizer(cfg, m_g.parameters())
stepper = GraphedTrainStep(m_g, cfg, optimizer=o_g)
assert stepper.dp and stepper.split_capture
for it in range(3):
    random.seed(10 + it)
    out = m_e(b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="train")
    lossf(*out, b["target_view"].unsqueeze(1), cfg)[0].backward()
    o_e.step()
    o_e.zero_grad()
    random.seed(10 + it)
    stepper(b["data"], b["input_theta"], b["target_theta"], b["rois"], b["target_view"])
torch.cuda.synchronize()
assert isinstance(next(iter(stepper.slots.values()))["graph"], tuple)          # the split capture was taken
pe = torch.cat([p.detach().reshape(-1) for p in m_e.parameters()])
pg = torch.cat([p.detach().reshape(-1) for p in m_g.parameters()])
assert torch.equal(pe, pg), float((pe - pg).abs().max())
dist.destroy_process_group()
print("RCCL_OK")
