"""Worker of test_data_parallel_plumbing_gloo_world2 (one process per rank, gloo)."""
import sys

import numpy as np
import torch
import torch.distributed as dist

rank, world, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dist.init_process_group("gloo", rank=rank, world_size=world)
from electrocardio_panorama_amd import parallel  # noqa: E402
from electrocardio_panorama_amd import synth  # noqa: E402

torch.manual_seed(0)
ws = [torch.randn(5, 3), torch.randn(7), torch.randn(2, 2, 2)]
full = synth.make_batch(8, 1, 64, seed=3)
shard = parallel.shard_batch(full, rank, world)
idx = parallel.shard_indices(8, rank, world)


def grads_for(batch):
    x = torch.from_numpy(batch["data"]).mean(dim=(1, 2))                     # [b]
    params = [w.clone().requires_grad_(True) for w in ws]
    loss = sum((p.sum() * x).mean() * (i + 1) for i, p in enumerate(params)) # mean over the (local) batch
    loss.backward()
    return params


local = grads_for(shard)
flat = torch.empty(sum(p.numel() for p in local))
parallel.reduce_flat_grads([p.grad for p in local], flat)                   # sum over ranks
flat *= 1.0 / world
expect = torch.cat([p.grad.reshape(-1) for p in grads_for(full)])
# the collective range check of Solver._check_h2_range: per-rank counters (rank 1 alone saw a clamp) become the same totals everywhere
counts = parallel.sum_counts([3 if rank == 1 else 0, rank], torch.device("cpu"))
np.savez(f"{out}/rank{rank}.npz", flat=flat.numpy(), expect=expect.numpy(), idx=np.asarray(idx), counts=np.asarray(counts))
dist.destroy_process_group()
