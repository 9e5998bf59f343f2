"""Per-kernel picture of one MU/FRO step at the NMFk sweep shape (65536 x 4096, small k): run under rocprofv3 --kernel-trace --stats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pydnmfk_amd.engine import HIP_OPS as ops
m, n = 65536, 4096
k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
norm = sys.argv[2] if len(sys.argv) > 2 else "fro"
dev = torch.device("cuda", 0)
A = torch.rand(m, n, device=dev); W = torch.rand(m, k, device=dev); H = torch.rand(k, n, device=dev)
for i in range(60):
    if norm == "fro": ops.mu_fro_step(A, W, H, 1.19e-7, True, i % 10 == 0)
    else: ops.mu_kl_step(A, W, H, 1.19e-7, True, i % 10 == 0)
torch.cuda.synchronize()
