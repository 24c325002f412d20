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
dist.destroy_process_group()
print("RCCL_OK")
