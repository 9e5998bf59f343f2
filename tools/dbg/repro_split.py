"""Bitwise run-to-run reproducibility of every bf16x6 product at sizes that fill the GPU (hazards show up as flipped bits)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pydnmfk_amd.engine import HIP_OPS_BF16X6 as x6
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
bad = 0
for (m, n, k, bf16) in [(65536, 8192, 64, False), (65536, 8192, 128, False), (65536, 8192, 64, True), (65536, 8192, 128, True),
                        (65536, 8192, 24, True), (32768, 16384, 16, False), (32768, 16384, 64, False), (32768, 16384, 128, False)]:
    A = torch.rand(m, n, device=dev, generator=g)
    if bf16:
        A = A.to(torch.bfloat16)
    W = torch.rand(m, k, device=dev, generator=g); H = torch.rand(k, n, device=dev, generator=g)
    names = ["aht", "wta"] + ([] if bf16 else ["kl_uht", "kl_wtu"])
    for name in names:
        outs = []
        for _ in range(4):
            shape = (m, k) if name in ("aht", "kl_uht") else (k, n)
            o = torch.empty(*shape, device=dev)
            if name == "aht": x6.aht(A, H, o)
            elif name == "wta": x6.wta(A, W, o)
            else: getattr(x6, name)(A, W, H, 1.19e-7, o)
            outs.append(o)
        ok = all(torch.equal(outs[0], o) for o in outs[1:]) and bool(torch.isfinite(outs[0]).all())
        bad += (not ok)
        print(m, n, k, "bf16" if bf16 else "f32", name, "ok" if ok else "DIFFERS")
    del A, W, H
print("failures:", bad)
