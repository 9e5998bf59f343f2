"""How much of a batched whole-fit call is GPU work?  wall time of ops.fit (B problems, itr steps) on a small shape; run under
`rocprofv3 --kernel-trace --stats` to get the sum of the kernel times next to it."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pydnmfk_amd.engine import HIP_OPS as ops, stack_alloc
m, n, k, B, itr = 1024, 256, int(os.environ.get("K", "8")), int(os.environ.get("B", "20")), int(os.environ.get("ITR", "100"))
method, norm = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("hals", "fro")
dev = torch.device("cuda")
A = stack_alloc(B, m, n, torch.float32 if (norm == "kl" or os.environ.get("DT") == "f32") else torch.bfloat16, dev); A.copy_(torch.rand(B, m, n, device=dev))
W = stack_alloc(B, m, k, torch.float32, dev); H = stack_alloc(B, k, n, torch.float32, dev)
out = {}
for rep in range(3):
    W.copy_(torch.rand(B, m, k, device=dev)); H.copy_(torch.rand(B, k, n, device=dev))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sq = ops.fit(method, norm, A, W, H, 1.19e-7, True, itr)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    out["rep%d" % rep] = {"issue_ms": (t1 - t0) * 1e3, "wall_ms": (t2 - t0) * 1e3}
print(json.dumps(out))
