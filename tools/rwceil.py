"""Read+write streaming ceilings of the GPU as seen through library / framework kernels that are not ours:
hipMemcpyAsync D2D, torch elementwise kernels (copy_, clamp_, mul_) on 1 GiB operands.  Bytes = reads + writes."""
import sys, os, json
import torch
dev = torch.device("cuda", 0)
n = 1 << 28                               # 1 GiB of fp32
x = torch.rand(n, device=dev); y = torch.empty_like(x); z = torch.rand(n, device=dev)
def t(fn, reps=20, warm=10):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    xs = sorted(s.elapsed_time(e) for s, e in ev)
    return xs[len(xs) // 2]
out = {}
for name, fn, nbytes in (("copy_ (1R+1W)", lambda: y.copy_(x), 8.0 * n),
                         ("clamp_ (1R+1W in place)", lambda: x.clamp_(min=1e-7), 8.0 * n),
                         ("mul_ (2R+1W in place)", lambda: x.mul_(z), 12.0 * n),
                         ("sum (1R)", lambda: x.sum(), 4.0 * n),
                         ("fill_ (1W)", lambda: y.fill_(1.0), 4.0 * n)):
    ms = t(fn); out[name] = {"ms": round(ms, 3), "GBs": round(nbytes / ms / 1e6, 1)}
print(json.dumps(out))
