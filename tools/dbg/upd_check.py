import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pydnmfk_amd.engine import HIP_OPS as ops, new_gram
EPS = 1.1920929e-07
dev = torch.device("cuda")
for k in (64, 32, 128):
    for m in (32, 64, 128, 4096, 89600):
        rs = np.random.RandomState(m + k)
        W = rs.rand(m, k).astype(np.float32); AH = (rs.rand(m, k) * 20).astype(np.float32)
        Hs = rs.rand(k, 96).astype(np.float32)
        G = ops.gram_hht(torch.from_numpy(Hs).to(dev), new_gram(k, dev))
        Wd = torch.from_numpy(W).to(dev)
        ops.mu_update_w(Wd, torch.from_numpy(AH).to(dev), G, EPS)
        ref = W.astype(np.float64) * (AH.astype(np.float64) / (W.astype(np.float64) @ G[:k, :k].cpu().numpy().astype(np.float64) + EPS))
        got = Wd.cpu().numpy()
        err = np.abs(got - ref) / (np.abs(ref) + 1e-12)
        bad = np.argwhere(err > 1e-4)
        print("W k=%d m=%d rel=%.2e bad=%d" % (k, m, np.linalg.norm(got - ref) / np.linalg.norm(ref), len(bad)),
              "rows", sorted(set(bad[:, 0].tolist()))[:12], "cols", sorted(set(bad[:, 1].tolist()))[:16])
        # H update with the same numbers transposed
        Hd = torch.from_numpy(np.ascontiguousarray(W.T)).to(dev)
        ops.mu_update_h(Hd, torch.from_numpy(np.ascontiguousarray(AH.T)).to(dev), G, EPS, False)
        Gn = G[:k, :k].cpu().numpy().astype(np.float64)
        refh = W.T.astype(np.float64) * (AH.T.astype(np.float64) / (Gn @ W.T.astype(np.float64) + EPS))
        goth = Hd.cpu().numpy()
        errh = np.abs(goth - refh) / (np.abs(refh) + 1e-12)
        badh = np.argwhere(errh > 1e-4)
        print("H k=%d n=%d rel=%.2e bad=%d" % (k, m, np.linalg.norm(goth - refh) / np.linalg.norm(refh), len(badh)),
              "rows", sorted(set(badh[:, 0].tolist()))[:12], "cols", sorted(set(badh[:, 1].tolist()))[:16])
