"""The reference's own training shape (batch 32, 3 leads, L = 512) as bench.py's `secondary` measures it: eager / replayed ms per step."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
r = bench.native_shape(torch.device("cuda:0"))
print(json.dumps({k: r[k] for k in ("eager", "graph") if k in r}))
