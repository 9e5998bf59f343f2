"""A/B of the pipelined U H^T kernel (csrc/dnmf_kluht.h) against kl_uht_kernel: run under the tuning build
(DNMF_LIB_PATH=tools/_build/libdnmf_hip_tune.so) once with DNMF_KLUHT_PIPE=1 and once with 0; each run writes its outputs
to <outdir>/uht_<tag>.pt, `compare` checks them bitwise and against float64.
  python tools/kluht_ab.py run <tag> <outdir> | compare <tagA> <tagB> <outdir>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

SHAPES = [(128, 64, 32), (256, 96, 64), (384, 2048, 128), (1000, 4096, 128), (4096 + 60, 8192, 64), (2048, 8192 + 32, 32),
          (640, 4096, 20), (512, 1024, 40), (896, 3072, 100), (8192, 16384, 128)]

def make(m, n, k, dev):
    g = torch.Generator(device=dev); g.manual_seed(m + 7 * n + 13 * k)
    A = torch.rand(m, n, device=dev, generator=g)
    A[A < 0.05] = 0.0                      # KL inputs carry exact zeros (swim)
    return A, torch.rand(m, k, device=dev, generator=g), torch.rand(k, n, device=dev, generator=g)

if sys.argv[1] == "run":
    from pydnmfk_amd.engine import HIP_OPS as ops
    dev = torch.device("cuda", 0)
    res = {}
    for (m, n, k) in SHAPES:
        A, W, H = make(m, n, k, dev)
        out = torch.empty(m, k, device=dev)
        ops.kl_uht(A, W, H, 1.19e-7, out)
        ref = ((A.double() / (W.double() @ H.double() + 1.19e-7)) @ H.double().t())
        err = ((out.double() - ref).norm() / ref.norm()).item()
        res[(m, n, k)] = out.cpu()
        print(m, n, k, "rel err vs float64 %.2e" % err, flush=True)
        assert err < 2e-6, err
    torch.save(res, os.path.join(sys.argv[3], "uht_%s.pt" % sys.argv[2]))
else:
    a = torch.load(os.path.join(sys.argv[4], "uht_%s.pt" % sys.argv[2]))
    b = torch.load(os.path.join(sys.argv[4], "uht_%s.pt" % sys.argv[3]))
    bad = 0
    for key in a:
        same = torch.equal(a[key], b[key])
        print(key, "bit identical" if same else "DIFFERENT max abs %.3e" % (a[key] - b[key]).abs().max().item())
        bad += not same
    sys.exit(1 if bad else 0)
