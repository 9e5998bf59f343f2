"""benchlib.common -- constants and helpers shared by bench.py and the per-config modules."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_CLOCK_GHZ = 2.4            # the clock that peak is quoted at (256 CUs x 256 flop per CU and clock)
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec


CONFIGS = {
    2: dict(m=65536, n=4096, k=32, norm="fro", steps=2000, warmup=20, label="BASELINE config 2"),
    3: dict(m=262144, n=8192, k=64, norm="fro", steps=500, warmup=5, label="BASELINE config 3"),
    4: dict(m=131072, n=65536, k=128, norm="kl", steps=30, warmup=3, label="BASELINE config 4"),
    5: dict(m=65536, n=4096, k=16, norm="fro", steps=1, warmup=1, label="BASELINE config 5"),
}


def flush_c_stdio():
    """RCCL 2.26 writes a version banner to C stdout when its first communicator comes up; on a pipe that text stays in the C
    buffer until the process exits, i.e. it would land AFTER rank 0's JSON line.  Flushing the C streams before the line is
    printed keeps the JSON the last line of stdout."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def event_time_ms(fn, reps=5, warm=2):
    import torch
    for _ in range(warm):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in evs:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return sum(ts) / len(ts), ts[0]


def pmc_traffic(role, workload="bench"):
    """HBM bytes per launch of a kernel from the NEWEST committed PMC pass of the given profiled workload
    (profiles/<tag>_<workload>_pmc.json: separate --pmc runs of tools/collect_profiles.sh, FETCH_SIZE doubled per
    MI355X_MICROARCH.md; older rounds: profiles/<tag>_pmc.json) -- a constant of the committed profile, not a live counter:
    the driver's run has no profiler attached.  Kernels are picked by role (template argument lists change between
    rounds): 'nt' = the k = 64 NT instantiation with the most bytes (the fused A H^T + W update; the gram launch only
    reads H), 'tn' = likewise for the TN form, otherwise a name prefix.  None if absent."""
    import glob
    best = None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")))
    files = [f for f in files if ("_%s_pmc" % workload) in f or os.path.basename(f).count("_") == 1]
    for f in files:
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        prefix = {"nt": "nt_kernel<2,", "tn": "tn_kernel<2,"}.get(role, role)
        cand = [(c["hbm_bytes"], name) for name, c in d.items()
                if name.startswith(prefix) and isinstance(c, dict) and "hbm_bytes" in c]
        if cand:
            by, name = max(cand)
            best = {"bytes": by, "source": os.path.basename(f), "kernel": name}
    return best


def rccl_record():
    """What the N > 1 line says about the transport: the RCCL the library bound (version code, where it was found) and the
    NCCL_* / RCCL_* environment in effect (algorithm / protocol overrides change what an allreduce of 2 MiB costs)."""
    rec = {"env": {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_")) or k == "HSA_ENABLE_IPC_MODE_LEGACY"}}
    try:
        import ctypes
        from pydnmfk_amd._lib import lib
        v, buf = ctypes.c_int(0), ctypes.create_string_buffer(256)
        if lib.dnmf_comm_rccl_version(ctypes.byref(v), buf, 256) == 0:
            rec["version_code"], rec["found"] = int(v.value), buf.value.decode()
    except Exception as exc:  # noqa: BLE001
        rec["version_error"] = repr(exc)
    try:
        import torch
        rec["torch_nccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:  # noqa: BLE001
        pass
    return rec


class Job:
    """What every configuration's timed region needs: the ranks, the barrier and the max-over-ranks clock of the contract."""

    def __init__(self, a, world, rank, local, dev, ctl, emu, rccl_ranks_seen):
        self.a, self.world, self.rank, self.local, self.dev, self.ctl = a, world, rank, local, dev, ctl
        self.emu, self.rccl_ranks_seen = emu, rccl_ranks_seen

    def barrier(self):
        import torch.distributed as dist
        if self.a.backend == "nccl":
            dist.barrier(device_ids=[self.local])
        else:
            dist.barrier()

    def max_over_ranks(self, x):
        import torch
        import torch.distributed as dist
        if self.world == 1:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=self.ctl)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def timed(self, nsteps, step):
        """EXACTLY nsteps steps between barrier + device sync on both sides; max over ranks."""
        import torch
        if self.world > 1:
            self.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(nsteps):
            step(i)
        torch.cuda.synchronize()
        if self.world > 1:
            self.barrier()
        torch.cuda.synchronize()
        return self.max_over_ranks(time.perf_counter() - t0)


def parse_grid(text, world, default):
    if not text:
        return default
    r, c = (int(x) for x in text.lower().split("x"))
    if r * c != world:
        sys.exit("bench.py: --grid %s needs %d ranks, the job has %d" % (text, r * c, world))
    return r, c
