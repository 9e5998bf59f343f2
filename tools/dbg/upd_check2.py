import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pydnmfk_amd.engine import HIP_OPS as ops, new_gram
EPS = 1.1920929e-07
dev = torch.device("cuda")
k, m = 128, 32
rs = np.random.RandomState(5)
W = rs.rand(m, k).astype(np.float32); AH = (rs.rand(m, k) * 20).astype(np.float32)
Hs = rs.rand(k, 96).astype(np.float32)
G = ops.gram_hht(torch.from_numpy(Hs).to(dev), new_gram(k, dev))
Gn = G[:k, :k].cpu().numpy().astype(np.float64)
print("G symmetric:", np.abs(Gn - Gn.T).max())
ref = W.astype(np.float64) * (AH.astype(np.float64) / (W.astype(np.float64) @ Gn + EPS))
for mode in ("aligned", "unaligned"):
    if mode == "aligned":
        Wd = torch.from_numpy(W).to(dev); Sd = torch.from_numpy(AH).to(dev)
    else:
        Wb = torch.zeros(m * k + 1, device=dev); Wb[1:] = torch.from_numpy(W).to(dev).reshape(-1); Wd = Wb[1:].view(m, k)
        Sb = torch.zeros(m * k + 1, device=dev); Sb[1:] = torch.from_numpy(AH).to(dev).reshape(-1); Sd = Sb[1:].view(m, k)
    ops.mu_update_w(Wd, Sd, G, EPS)
    got = Wd.cpu().numpy()
    err = np.abs(got - ref) / (np.abs(ref) + 1e-12)
    bad = np.argwhere(err > 1e-4)
    print(mode, "bad", len(bad))
    for (i, j) in bad[:10]:
        print("  ", i, j, "got", got[i, j], "ref", ref[i, j], "W0", W[i, j], "S", AH[i, j], "den", (W.astype(np.float64) @ Gn)[i, j])
    # which S value would explain it?  got = W0 * S' / den
    if len(bad):
        i, j = bad[0]
        sp = got[i, j] * (W.astype(np.float64) @ Gn)[i, j] / W[i, j]
        print("   implied S'", sp, "nearest S col", int(np.argmin(np.abs(AH[i] - sp))), "nearest in any row", np.argwhere(np.abs(AH - sp) < 1e-3)[:4].tolist())
