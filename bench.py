#!/usr/bin/env python3
"""bench.py -- MU iterations/sec of the MI355X update engine on BASELINE.json's headline workload.

Workload (config.workload): synthetic dense fp32 X = 262144 x 8192, k = 64, MU / Frobenius, 1D row grid
p_r = N, p_c = 1 (BASELINE config 3).  The global X is fixed, so N GPUs each hold m/N rows: strong scaling.
A "step" is one full MU iteration exactly as PyNMF.fit runs it (W update, H update with the new W, the
allreduce of [W^T A | W^T W] when N > 1, and the clamp on every 10th step), X already resident in HBM.

  python bench.py                      # N=1
  python bench.py --gpus N             # N>1 typed by hand: starts its own N ranks (fresh child processes) and relays
                                       # rank 0's JSON line and exit code
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W        # what the driver runs

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline     : the dominant kernel (fused A H^T + W update, one launch), algorithmic flops / HIP-event time
  rooflines    : the same for every big kernel (fused NT, TN, residual) and the HBM-bound class
  roofline_hbm : the eltwise multiply-divide kernel (dnmf_mu_update_h) in isolation on SURVEY 8d's shape (k x 2^22,
                 12 B per element = 3.2 GB at k = 64), beside it the norm kernel on X
  kernels      : per-kernel HIP-event times measured in situ (the step replayed primitive by primitive)
  sustained    : when the timed region asked for is shorter than 2 s, the same step timed again over >= 300 steps
                 (its own ms/step: shows the clock the chip HOLDS under the load, not its boost)
  multi_gpu    : (N > 1) rccl_ranks_seen, per-rank compute-only ms (the same step with the exchange stubbed out) next to
                 the full step -> exposed_comm_ms, the packed allreduce timed alone, and an A/B of the exchange transport
                 (torch.distributed vs the library's own RCCL communicator) x the H phase's overlap chunking (1 / 2 / 4)
                 measured in the warm-up; the fastest pair is used for the timed region
  cpu_baseline : the numpy oracle (port of the reference's path) in the reference's process model -- P = min(8, cores)
                 single-thread processes, each on its 1/P row slab of X -- timed on the host (N=1 only)
With no flags: N = 1 and 500 timed steps (about 2.3 s of GPU time).

--config {2,3,4,5} (round 4) selects the BASELINE configuration that is measured; 3 (the headline) is the default and is
unchanged.  2 = MU/FRO 65536 x 4096, k = 32 (the same step on the smaller problem).  4 = MU/KL 131072 x 65536, k = 128 on the
2D grid the rank count gives (8 ranks: 4 x 2) through nmf_algorithms_2D -- `--emulate-ranks 8` runs ONE rank's 32768 x 32768
block on one GPU with the real collective calls on one-rank groups; roofline = the KL W-side product.  5 = the NMFk sweep
k = 2..16 x 20 perturbations, HALS/FRO on bf16-stored X (a "step" is one whole sweep; value = fits per second).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib.common import (CONFIGS, PEAK_CLOCK_GHZ, PEAK_FP32_MFMA_TFLOPS, PEAK_HBM_GBS, Job, event_time_ms, flush_c_stdio,  # noqa: E402
                             parse_grid, pmc_traffic, rccl_record)
from benchlib.config4 import run_config4  # noqa: E402
from benchlib.config5 import run_config5, swim_example  # noqa: E402
from benchlib.cpu_baseline import (_cpu_hals_rank, _cpu_kl_rank, _cpu_pool, cpu_baseline, cpu_baseline_allcores,  # noqa: E402
                                   host_cpu)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 4, 5],
                    help="BASELINE.json configuration: 3 = the headline (default); 2 = MU/FRO 65536x4096 k=32; 4 = MU/KL 131072x65536 "
                         "k=128 on the 2D grid; 5 = NMFk sweep k=2..16 x 20 perturbations, HALS/FRO, bf16-stored X")
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default per config: 500 / 2000 / 30 / 1)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--rows", dest="m", type=int, default=None)
    ap.add_argument("--cols", dest="n", type=int, default=None)
    ap.add_argument("--rank", dest="k", type=int, default=None)
    ap.add_argument("--norm", default=None)
    ap.add_argument("--grid", default=None, help="config 4 / 5: process grid 'RxC' (default: 8 ranks 4x2, 4: 2x2, 2: 2x1 for config 4; 1xN for config 5)")
    ap.add_argument("--perturbations", type=int, default=20, help="config 5")
    ap.add_argument("--itr", type=int, default=100, help="config 5: HALS iterations per fit")
    ap.add_argument("--nmfk-batch", default="auto", help="config 5: perturbation fits per batched whole-fit call (auto = all that fit in memory; "
                    "1 = one by one, each still ONE library call; with --fit-loop python = the round-4 path: a Python step loop per fit)")
    ap.add_argument("--fit-loop", default="native", choices=["native", "python"], help="config 5: whole fits in the library (default) or the per-step Python loop")
    ap.add_argument("--start-k", type=int, default=2, help="config 5")
    ap.add_argument("--end-k", type=int, default=16, help="config 5")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for debugging)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-sustained", action="store_true")
    ap.add_argument("--gemm", default="fp32", choices=["fp32", "bf16x6"],
                    help="arithmetic of A.H^T and W^T.A in the measured step: fp32 MFMA (default, the parity reference) or six "
                         "bf16 piece products per fp32 product (fp32-grade, csrc/dnmf_split.h)")
    ap.add_argument("--no-bf16x6", action="store_true", help="skip the extra bf16x6 measurement of the default run")
    ap.add_argument("--onepass", default="auto", choices=["auto", "on", "off"], help="MU/FRO at 16 < k <= 32 on one rank: the one-pass team kernel where it "
                    "measured faster (auto, the library's default), wherever the shape allows it (on), never (off) -- dnmf_set_onepass")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configurations (2, 4 on one emulated rank, 5 at the example size) "
                    "that the default run measures after the headline and reports in `configs`")
    ap.add_argument("--no-swim", action="store_true", help="skip the extra measurement of the reference's own example sweep (swim, KL / MU NMFk) in the default run")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="diagnostic, single GPU: run ONE rank's share of an R-GPU row grid (m / R rows) with the REAL exchange calls on "
                         "a one-rank group (RCCL has no wire to cross: kernels, launch overheads and the collectives' fixed costs are "
                         "real, the transfer is not) -- exercises the whole N > 1 path of this program, A/B included, on one GPU")
    ap.add_argument("--exchange", default="auto", choices=["auto", "torch", "native"],
                    help="N > 1: who issues the allreduce -- torch.distributed between the kernel launches, or the library's own "
                         "RCCL communicator inside a one-call step (csrc/dnmf_comm.hip); 'auto' (default) times both in the "
                         "warm-up and uses the faster")
    ap.add_argument("--overlap-2d", action="store_true", help="config 4 on a 2D grid (torch-sequenced exchange): start the allgather of the updated "
                    "H slices behind each step's H update (params.overlap_2d), so that it runs under the next step's first kernels")
    ap.add_argument("--overlap-chunks", default="auto",
                    help="N > 1: column chunks of the H phase's overlapped exchange; 'auto' (default) times 1 / 2 / 4 in "
                         "the warm-up and uses the fastest")
    a = ap.parse_args()
    dflt = CONFIGS[a.config]
    for key in ("m", "n", "k", "norm", "steps", "warmup"):
        if getattr(a, key) is None:
            setattr(a, key, dflt[key])
    return a


# BASELINE.json `configs` (1 = the reference's own CPU case: a parity test, tests/test_gpu_parity.py, not a bench line)


def other_configs():
    """The other BASELINE configurations next to the headline, each as a CHILD process of this one (a fresh `python bench.py --config N`:
    its own context, its own clocks; this process only waits) with a bounded workload:
      2  MU/FRO 65536 x 4096, k = 32, 1000 steps;
      4  MU/KL, ONE rank's 32768 x 32768 block of the 4 x 2 grid with the real collective calls on one-rank groups (--emulate-ranks 8;
         not a whole-job number: the workload string says so), 10 steps;
      5  the NMFk sweep k = 2..16 x 20 perturbations, HALS/FRO on bf16-stored X at the reference's example size 1024 x 256, one sweep.
    Returns {name: {metric, value, unit, ms_per_step, steps, config.workload, roofline{kernel, bound, achieved, peak, unit, frac}}}; a child
    that fails or takes more than two minutes is reported as {"error": ...} and never costs the headline line."""
    import subprocess
    runs = {
        "config2": ["--config", "2", "--steps", "1000", "--warmup", "20", "--no-cpu-baseline", "--no-sustained", "--no-bf16x6", "--no-swim"],
        "config4_one_emulated_rank": ["--config", "4", "--emulate-ranks", "8", "--steps", "10", "--warmup", "2", "--no-cpu-baseline"],
        "config5_example_size": ["--config", "5", "--rows", "1024", "--cols", "256", "--no-cpu-baseline", "--no-kernel-timing"],
    }
    env = dict(os.environ)
    for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    out = {}
    for name, args in runs.items():
        t0 = time.time()
        try:
            res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-configs"] + args, capture_output=True, text=True,
                                 timeout=120, env=env)
            line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
            if res.returncode != 0 or not line:
                out[name] = {"error": ("exit %d: " % res.returncode) + (res.stderr.strip().splitlines() or [""])[-1][:200]}
                continue
            d = json.loads(line[-1])
            e = {key: d.get(key) for key in ("metric", "value", "unit", "ms_per_step", "steps", "dtype")}
            e["config"] = {"workload": d.get("config", {}).get("workload")}
            r = d.get("roofline")
            if r:
                e["roofline"] = {key: r.get(key) for key in ("kernel", "bound", "achieved", "peak", "unit", "frac", "ms_per_launch")}
            for key in ("estimated_k", "fits_per_step", "onepass"):
                if key in d:
                    e[key] = d[key]
            e["wall_s"] = round(time.time() - t0, 1)
            out[name] = e
        except Exception as ex:  # noqa: BLE001
            out[name] = {"error": str(ex)[:200]}
    return out


def launch_ranks(a):
    """`python bench.py --gpus N` typed by hand (no launcher environment): start N fresh ranks as CHILD processes through
    torch.distributed.run, relay rank 0's JSON line and the exit code.  Nothing in this parent touches the GPU: the device
    count comes from the KFD topology in sysfs (no HIP call), and the parent never re-executes itself."""
    import socket
    import subprocess
    ndev = count_gpus()
    oversub = bool(os.environ.get("DNMF_BENCH_OVERSUBSCRIBE")) and a.backend != "nccl"
    if ndev < a.gpus and not oversub:
        sys.stderr.write("bench.py: --gpus %d but this node has %d GPU(s) (DNMF_BENCH_OVERSUBSCRIBE=1 with --backend gloo "
                         "stacks ranks on one device for debugging)\n" % (a.gpus, ndev))
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, env=env)                    # stdout / stderr are inherited: rank 0's line goes straight out
    try:
        rc = proc.wait()
    except KeyboardInterrupt:
        proc.terminate()
        rc = 130
    if rc != 0:
        sys.stderr.write("bench.py: the %d-rank launch exited with code %d\n" % (a.gpus, rc))
    return rc


def count_gpus():
    """GPUs of this node without initialising HIP: KFD topology nodes with SIMDs (CPUs have simd_count 0); falls back to
    torch.cuda.device_count() (which does not create a context on this stack) when sysfs is not readable."""
    import glob
    n, seen = 0, False
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            for line in open(f):
                if line.startswith("simd_count"):
                    seen = True
                    n += int(line.split()[1]) > 0
        except (OSError, ValueError):
            pass
    if seen:
        vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
        if vis:
            n = min(n, len([v for v in vis.split(",") if v.strip() != ""]))
        return n
    import torch
    return torch.cuda.device_count()


def main():
    a = parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        a.gpus = world                          # the launcher's rank count is authoritative
    ndev = torch.cuda.device_count()
    if world > ndev and not (a.backend != "nccl" and os.environ.get("DNMF_BENCH_OVERSUBSCRIBE")):
        sys.exit("bench.py: %d ranks but %d GPUs (RCCL needs one GPU per rank)" % (world, ndev))
    local = local % max(1, ndev)            # gloo debugging runs may stack ranks on one device
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ctl = dev if a.backend == "nccl" else torch.device("cpu")      # where control-plane scalars of this bench are reduced
    rccl_ranks_seen = None
    emu = a.emulate_ranks if (world == 1 and a.emulate_ranks > 1) else 0
    if emu:
        import socket
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(sk.getsockname()[1])
        sk.close()
        dist.init_process_group(a.backend, rank=0, world_size=1)
        from pydnmfk_amd.dist_comm import TorchComm
        TorchComm.always_collective = True      # the one-rank communicator still issues its collectives
        rccl_ranks_seen = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(a.backend)      # nccl = RCCL; communicators are created lazily on the current device
        # one tiny collective up front: every rank contributes 1, so the sum is the number of ranks the transport really
        # connects.  If RCCL cannot come up the run ends here with a non-zero exit (the launcher then stops the other
        # ranks).  There is no per-rank fallback to another transport: ranks deciding that on their own could end up in
        # different process groups.  `--backend gloo` selects the host-staged transport explicitly (debugging only;
        # labelled in config.parallelism).
        try:
            t = torch.ones(1, device=ctl)
            dist.all_reduce(t)
            if ctl.type == "cuda":
                torch.cuda.synchronize()
            rccl_ranks_seen = int(t.item())
            if rccl_ranks_seen != world:
                raise RuntimeError("allreduce of ones returned %s, expected %d" % (t.item(), world))
        except Exception as exc:  # noqa: BLE001
            sys.stderr.write("bench.py: %s did not come up on rank %d (%s)\n" % (a.backend, rank, exc))
            sys.stderr.flush()
            os._exit(3)

    if a.config in (4, 5):
        job = Job(a, world, rank, local, dev, ctl, emu, rccl_ranks_seen)
        out = run_config4(a, job) if a.config == 4 else run_config5(a, job)
        if rank == 0:
            flush_c_stdio()
            print(json.dumps(out), flush=True)
        if world > 1:
            job.barrier()
        if world > 1 or emu:
            dist.destroy_process_group()
        return

    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.dist_nmf import nmf_algorithms_1D
    from pydnmfk_amd.engine import new_gram, ops_for
    from pydnmfk_amd.utils import determine_block_params, parse

    m, n, k = a.m, a.n, a.k
    from pydnmfk_amd._lib import lib as _lib
    _lib.dnmf_set_onepass({"auto": 1, "on": 2, "off": 0}[a.onepass])
    comms = MPI_comm(None, world, 1)
    grid_r = emu or world                       # rows of the process grid the step is written for
    multi = world > 1 or emu > 1
    p = parse()
    p.comm1, p.comm, p.p_r, p.p_c, p.k, p.m, p.n = comms.comm, comms, grid_r, 1, k, m, n
    p.row_comm, p.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
    p.norm, p.method, p.W_update, p.eps = a.norm, "mu", True, 1.1920929e-07
    p.gemm = a.gemm
    ops = ops_for(p)
    m_l = determine_block_params(rank, (grid_r, 1), (m, n)).determine_block_shape_asymm()[0]

    # synthetic data, generated on device: X ~ U[0,1), W0 per rank, H0 from rank 0
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    A = torch.rand(m_l, n, device=dev, generator=g)
    g.manual_seed(4321 + rank)
    W = torch.rand(m_l, k, device=dev, generator=g)
    g.manual_seed(99)
    H = torch.rand(k, n, device=dev, generator=g)
    if world > 1:
        H = comms.comm.bcast(H, root=0)

    def barrier():
        if a.backend == "nccl":
            dist.barrier(device_ids=[local])
        else:
            dist.barrier()

    def step(i):
        nmf_algorithms_1D(A, W, H, params=p).update(clamp=(i % 10 == 0))

    p6 = parse()
    p6.__dict__.update(vars(p))
    p6.gemm = "bf16x6"

    def step_x6(i):
        nmf_algorithms_1D(A, W, H, params=p6).update(clamp=(i % 10 == 0))

    def max_over_ranks(x):
        if world == 1:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=ctl)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def timed(nsteps, step=step):
        """EXACTLY nsteps steps between barrier + device sync on both sides; max over ranks."""
        if world > 1:
            barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(nsteps):
            step(i)
        torch.cuda.synchronize()
        if world > 1:
            barrier()
        torch.cuda.synchronize()
        return max_over_ranks(time.perf_counter() - t0)

    # N > 1, before the timed region (untimed, part of the warm-up): how many column chunks the H phase's exchange is cut
    # into (dist_nmf._fro_h_phase_overlapped: W^T A of chunk c+1 is computed while chunk c is being reduced) is decided from
    # DATA -- 1 (one packed allreduce), 2 and 4 chunks are each timed over a few steps, every rank sees the same
    # max-over-ranks times and therefore picks the same winner.
    mg = None
    if multi:
        mg = {"rccl_ranks_seen": rccl_ranks_seen, "backend": a.backend, "rccl": rccl_record() if a.backend == "nccl" else None}
        if emu:
            mg["emulated"] = ("ONE rank's share of a %d-GPU row grid on a single GPU: real kernels, launches and collective calls on a "
                              "one-rank group, no wire time" % emu)
        # exchange transports: torch.distributed (dist.all_reduce between the kernel launches) and, over RCCL, the
        # library's own communicator, where a whole step -- kernels, allreduce, kernels -- is ONE C call (no Python between
        # the launches; csrc/dnmf_comm.hip).  Both run the same kernels in the same order.
        modes = ["torch"] if a.exchange in ("auto", "torch") else []
        hosted = a.backend != "nccl" and bool(os.environ.get("DNMF_BENCH_HOSTED"))   # (tests: the library-sequenced arms over gloo,
        if (a.backend == "nccl" or hosted) and a.exchange in ("auto", "native") and a.gemm == "fp32":   #  ranks stacked on one GPU)
            ok, why = 1, None
            try:
                from pydnmfk_amd.engine import NativeComm, _torch_hosted_collective
                if hosted:
                    p._native_comm = NativeComm.hosted(world, rank, world, 1, _torch_hosted_collective(
                        {0: comms.comm, 1: comms.cart_1d_row(), 2: comms.cart_1d_column()}))
                    mg["native_transport"] = "hosted over " + a.backend
                else:
                    p._native_comm = NativeComm(comms.comm, world, 1)
                if emu:                                            # a one-rank communicator that still issues its RCCL calls
                    p._native_comm.set_always_exchange(True)
                    p.native_always = True
                t1 = torch.ones(4, device=dev)                     # smoke: the library's communicator reduces over every rank
                p._native_comm.allreduce_(t1)
                torch.cuda.synchronize()
                if int(t1[0].item()) != world:
                    raise RuntimeError("library allreduce of ones returned %s, expected %d" % (t1[0].item(), world))
            except Exception as exc:  # noqa: BLE001
                ok, why = 0, repr(exc)
            ok = int(-max_over_ranks(-float(ok)))                  # usable only if every rank has it
            if ok:
                modes.append("native")
                # third arm: the direct two-shot allreduce over IPC peer buffers for the packed
                # message -- eligible only if every rank connected AND its sum agrees with RCCL's on a test vector here
                # (DNMF_BENCH_NO_DIRECT=1 leaves this arm out)
                if world > 1 and a.exchange == "auto" and not os.environ.get("DNMF_BENCH_NO_DIRECT"):
                    try:
                        kp_ = 32 if k <= 32 else (64 if k <= 64 else 128)
                        # sized from GLOBAL quantities: every rank must ask for the same capacity (the asymmetric m_l differs between ranks)
                        nmsg = max(k * n, -(-m // world) * k) + 8 * 64 + kp_ * kp_ + 1024
                        good = p._native_comm.enable_direct(comms.comm, nmsg)
                        if good:
                            xt = torch.rand(k * n + kp_ * kp_, device=dev) - 0.25
                            y_ring = p._native_comm.allreduce_(xt.clone())
                            y_dir = p._native_comm.allreduce_direct_(xt.clone())
                            torch.cuda.synchronize()
                            err_d = float((y_dir - y_ring).abs().max() / y_ring.abs().max())
                            good = err_d < 1e-5 and not p._native_comm.direct_timed_out()
                            mg["direct_vs_rccl_max_rel_diff"] = err_d
                        good = int(-max_over_ranks(-float(bool(good))))
                    except Exception as exc:  # noqa: BLE001
                        good = 0
                        mg["direct_exchange_unavailable"] = repr(exc)
                        good = int(-max_over_ranks(-float(good)))
                    if good:
                        modes.append("native-direct")
                    else:
                        mg.setdefault("direct_exchange_unavailable", "not connected on every rank, or its sum disagreed with RCCL's")
            else:
                mg["native_exchange_unavailable"] = why or "another rank could not create the library communicator"
        if not modes:
            sys.exit("bench.py: --exchange %s is not available (backend %s)" % (a.exchange, a.backend))
        chunks = (1, 2, 4) if (a.overlap_chunks == "auto" and a.norm == "fro") else ((1,) if a.overlap_chunks == "auto" else (int(a.overlap_chunks),))
        if len(modes) * len(chunks) > 1:
            ab, nab = {}, max(5, min(40, a.steps))
            for mode in modes:
                p.exchange = "native" if mode.startswith("native") else mode
                if mode.startswith("native"):
                    p._native_comm.set_direct(mode == "native-direct")
                for c in chunks:
                    p.overlap_chunks = c
                    for i in range(3):
                        step(i)
                    ab[(mode, c)] = timed(nab) / nab * 1e3
            best = min(ab, key=ab.get)
            mg["exchange_ab_ms_per_step"] = {"%s/chunks=%d" % kc: v for kc, v in ab.items()}
            mg["exchange_ab_steps"] = nab
        else:
            best = (modes[0], chunks[0])
        p.exchange, p.overlap_chunks = ("native" if best[0].startswith("native") else best[0]), best[1]
        if best[0].startswith("native"):
            p._native_comm.set_direct(best[0] == "native-direct")
        mg["exchange_used"], mg["overlap_chunks_used"] = best
        p6.overlap_chunks = p.overlap_chunks

    for i in range(a.warmup):
        step(i)
    if world > 1:
        barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        barrier()
    torch.cuda.synchronize()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    assert torch.isfinite(W).all() and torch.isfinite(H).all()
    if multi and mg.get("exchange_used") == "native-direct":
        # a direct allreduce that gave up on a peer inside the timed loop would have timed garbage: fatal on every rank together
        if max_over_ranks(1.0 if p._native_comm.direct_timed_out() else 0.0) > 0:
            sys.exit("bench.py: a direct allreduce timed out during the timed steps -- the measurement is void")

    # N > 1: where the step's time goes.  (a) the same step with the exchange stubbed out (NullExchange: every allreduce
    # returns at once, so H diverges between ranks -- timing only; H is re-broadcast afterwards): per-rank compute-only
    # ms; exposed_comm_ms = full step - slowest rank's compute-only step.  (b) the packed [W^T A | W^T W] allreduce alone.
    if multi:
        from pydnmfk_amd.dist_comm import NullExchange
        pn = parse()
        pn.__dict__.update(vars(p))
        native_used = getattr(p, "exchange", "torch") == "native"
        if native_used:
            p._native_comm.set_null_exchange(True)                 # same one-call step, RCCL calls skipped
        else:
            pn.comm1 = NullExchange(p.comm1)                       # same Python choreography, exchanges return at once
        H_keep = H.clone()

        def step_nocomm(i):
            nmf_algorithms_1D(A, W, H, params=pn).update(clamp=(i % 10 == 0))

        for i in range(3):
            step_nocomm(i)
        nn = max(10, min(200, a.steps))
        if world > 1:
            barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(nn):
            step_nocomm(i)
        torch.cuda.synchronize()
        mine = (time.perf_counter() - t0) / nn * 1e3
        if native_used:
            p._native_comm.set_null_exchange(False)
        per = [None] * world
        dist.all_gather_object(per, mine)
        H.copy_(H_keep)
        H.copy_(comms.comm.bcast(H, root=0))
        del H_keep
        kp_ = 32 if k <= 32 else (64 if k <= 64 else 128)
        xbuf = torch.rand(k * n + kp_ * kp_, device=dev)
        for _ in range(3):
            comms.comm.allreduce_(xbuf)
        nx = 20
        if world > 1:
            barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nx):
            comms.comm.allreduce_(xbuf)
        torch.cuda.synchronize()
        t_ar = max_over_ranks((time.perf_counter() - t0) / nx * 1e3)
        del xbuf
        mg["compute_only_ms_per_rank"] = per
        mg["compute_only_ms"] = max(per)
        mg["full_step_ms"] = elapsed / a.steps * 1e3
        mg["exposed_comm_ms"] = elapsed / a.steps * 1e3 - max(per)
        mg["allreduce_alone_ms"] = t_ar
        mg["allreduce_bytes"] = 4 * (k * n + kp_ * kp_)
        mg["note"] = ("compute_only = the same step on the same transport with every exchange stubbed out (timing only); exposed_comm_ms = full "
                      "step - slowest rank's compute-only step; allreduce_alone = the packed [W^T A | W^T W] message "
                      "reduced back to back with nothing else on the GPU")

    out = None
    if rank == 0:
        ms = elapsed / a.steps * 1e3
        flops_iter = 4.0 * m * n * k + 4.0 * (m + n) * k * k      # SURVEY 8d, whole job
        out = {
            "metric": "mu_iterations_per_sec", "value": a.steps / elapsed, "unit": "iter/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if a.gemm == "fp32" else "f32 operands as 3 bf16 pieces, 6 bf16 MFMA products per fp32 product, fp32 accumulation",
            "data": "synthetic",
            "config": {"workload": "MU/%s X=%dx%d fp32 k=%d, 1D row grid p_r=%d p_c=1 (%s)%s" % (
                a.norm.upper(), m, n, k, grid_r, CONFIGS[a.config]["label"], " -- EMULATED: one rank's share on one GPU, not a whole-job number" if emu else ""), "m": m, "n": n, "k": k, "rows_per_gpu": m_l, "gemm": a.gemm,
                "parallelism": ("row-sharded X, allreduce[W^T A | W^T W] over %s" % ("RCCL" if a.backend == "nccl" else a.backend + " (host staged)")) if world > 1 else "single GPU"},
            "step_tflops_per_gpu": flops_iter / world / (ms * 1e-3) / 1e12,
            "step_mfma_frac": (flops_iter / world / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS) if a.gemm == "fp32" else None,
            "step_algorithmic_hbm_gbs_per_gpu": (4.0 * m_l * n + 12.0 * (m_l + n) * k) / (ms * 1e-3) / 1e9,
        }
        if not multi and a.gemm == "fp32":
            out["onepass"] = bool(_lib.dnmf_mu_fro_onepass(m_l, n, k))   # the step read A once (csrc/dnmf_team.h) instead of twice
        if mg is not None:
            out["multi_gpu"] = mg

    # A timed region of a few tens of ms (the driver's --steps 20) shows the boost clock; the big kernels are bound by the
    # clock the chip HOLDS under MFMA + HBM load.  Time the same step again over >= 300 steps (every rank takes part).
    if not a.no_sustained and elapsed < 2.0:
        ns = max(300, int(2.5 / max(elapsed / a.steps, 1e-6)))
        ns = min(ns, 20000)
        el = timed(ns)
        if rank == 0:
            out["sustained"] = {"steps": ns, "seconds": el, "ms_per_step": el / ns * 1e3, "value": ns / el,
                                "note": "same step, longer timed region (the headline value above is the K steps asked for)"}
            out["sustained_ms_per_step"] = el / ns * 1e3

    # Which shader clock does the GPU hold under this step?  A one-wave probe (dnmf_clock_probe) samples s_memtime against the
    # 100 MHz wall clock on a stream of its own while the same steps run once more, untimed.  The roofline fractions in this
    # line are quoted against the 2.4 GHz peak clock; the fp32 passes hold about 2.0 GHz (full-rate fp32 MFMAs + the HBM stream
    # meet the board's power limit -- DESIGN.md section 3), so frac = (matrix-pipe busy) x (held / peak clock).
    if not multi and rank == 0 and not a.no_kernel_timing:
        from pydnmfk_amd.engine import ClockProbe
        ms_step = elapsed / a.steps * 1e3

        def held_clock(fn, window_ms=60.0):
            nrep = max(3, int(window_ms / ms_step) + 1)
            torch.cuda.synchronize()
            pr = ClockProbe(nrep * ms_step * 1.3 + 2.0, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(nrep):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            dur = e0.elapsed_time(e1)
            g = pr.held_ghz(0.3 * dur, 0.95 * dur)                # the clock needs a few ms to settle; the tail is idle
            return None if g is None else round(g, 3)

        try:
            hc = held_clock(step)
        except Exception as ex:                                   # a measurement aid must not take the benchmark line down
            hc = None
            out["held_clock_note"] = "clock probe failed: %s" % ex

            def held_clock(fn, window_ms=60.0):
                return None
        out["held_clock_ghz"] = hc
        out["peak_clock_ghz"] = PEAK_CLOCK_GHZ
        if hc and out.get("step_mfma_frac") is not None:
            out["step_mfma_frac_at_held_clock"] = out["step_mfma_frac"] * PEAK_CLOCK_GHZ / hc

    # The same step with the two big contractions as six bf16 piece products per fp32 product (opt-in, params.gemm =
    # 'bf16x6'): reported NEXT TO the fp32-MFMA headline, never as it.  Same factors, same data, every rank takes part.
    if not multi and a.gemm == "fp32" and not a.no_bf16x6 and a.norm == "fro" and 32 < k <= 128 and n % 128 == 0:
        for i in range(5):
            step_x6(i)
        ns6 = max(100, int(1.5 / max(elapsed / a.steps, 1e-6)))
        el6 = timed(ns6, step_x6)
        if rank == 0:
            out["bf16x6"] = {
                "value": ns6 / el6, "unit": "iter/s", "ms_per_step": el6 / ns6 * 1e3, "steps": ns6,
                "held_clock_ghz": held_clock(step_x6) if not a.no_kernel_timing else None,
                "speedup_vs_fp32_mfma": (ns6 / el6) / (out["sustained"]["value"] if "sustained" in out else out["value"]),
                "step_algorithmic_hbm_gbs_per_gpu": (8.0 * m_l * n + 12.0 * (m_l + n) * k) / (el6 / ns6) / 1e9,   # X is read twice per step
                "note": "opt-in arithmetic (params.gemm='bf16x6', bench.py --gemm bf16x6): fp32 operands cut into three bf16 "
                        "pieces, six bf16 MFMA products per fp32 product, fp32 accumulation; as close to float64 as the fp32 "
                        "MFMA path (tests/test_gpu_split.py); the two passes over X become HBM bound"}

    # Informational: the same problem with X STORED as bfloat16 (params.precision = 'bfloat16': X is rounded once, arithmetic and
    # factors stay fp32) -- with the fp32-MFMA kernels (no gain at this rank: they are matrix-pipe bound) and with bf16x6, where
    # X is its own single piece and a product is three bf16 MFMAs.  A different X than the headline's, hence its own key.
    if not multi and a.gemm == "fp32" and not a.no_bf16x6 and a.norm == "fro" and 32 < k <= 128 and n % 128 == 0:
        A_f32 = A
        A = A_f32.to(torch.bfloat16)
        res16 = {}
        for name, fn in (("fp32_mfma", step), ("bf16x6", step_x6)):
            for i in range(3):
                fn(i)
            nsb = max(50, int(0.7 / max(elapsed / a.steps, 1e-6)))
            elb = timed(nsb, fn)
            res16[name] = {"value": nsb / elb, "unit": "iter/s", "ms_per_step": elb / nsb * 1e3, "steps": nsb}
        A = A_f32
        del A_f32
        if rank == 0:
            res16["note"] = ("X stored as bfloat16 (half the HBM bytes; a rounded X, not the headline's fp32 X); fp32 factors and "
                             "accumulation; bf16x6 = X times the three bf16 pieces of the fp32 factor")
            out["bf16_stored_x"] = res16

    if not a.no_kernel_timing and a.norm == "fro":
        # Per-kernel HIP-event timings IN SITU: the step is replayed primitive by primitive (same launches, same order
        # as dnmf_mu_fro_step / the 1D-row choreography) with an event pair around every library call, so each kernel
        # runs in the cache / clock state it sees inside the real step.  Events are recorded on torch's current stream,
        # which is the stream every launch goes to.  (The event pairs cost ~2-3 %: the per-kernel times sum to slightly
        # more than ms_per_step.)
        G = new_gram(k, dev)
        AtW = torch.empty(k, n, device=dev)
        reps = 10
        calls = [("gram_hht", lambda: ops.gram_hht(H, G)),
                 ("aht_update_w", lambda: ops.aht_update_w(A, H, G, W, p.eps)),
                 ("gram_wtw", lambda: ops.gram_wtw(W, G)),
                 ("wta", lambda: ops.wta(A, W, AtW)),
                 ("mu_update_h", lambda: ops.mu_update_h(H, AtW, G, p.eps, False))]
        evs = {name: [] for name, _ in calls}
        for it in range(reps + 2):
            for name, fn in calls:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                fn()
                e.record()
                if it >= 2:
                    evs[name].append((s, e))
        torch.cuda.synchronize()
        t = {name: sum(s.elapsed_time(e) for s, e in v) / len(v) for name, v in evs.items()}
        x6_t = None
        if 32 < k <= 128 and n % 128 == 0:
            ops6 = ops_for(p6)
            x6_t = {"aht_update_w": event_time_ms(lambda: ops6.aht_update_w(A, H, G, W, p.eps), reps=10, warm=3)[0],
                    "wta": event_time_ms(lambda: ops6.wta(A, W, AtW), reps=10, warm=3)[0]}
        t_sq, _ = event_time_ms(lambda: ops.sqnorm(A), reps=10, warm=3)
        t_res, _ = event_time_ms(lambda: ops.resid_sqnorm(A, W, H), reps=10, warm=3)
        # the eltwise multiply-divide kernel in isolation (measure it on a large n; k x 2^22 = 3.2 GB at
        # k = 64): in the step itself H is 2 MiB and the kernel is a 15 us latency chain
        n_iso = 1 << 22
        Hi = torch.rand(k, n_iso, device=dev)
        Si = torch.rand(k, n_iso, device=dev)
        Gi = new_gram(k, dev)
        Gi[:k, :k] = torch.rand(k, k, device=dev) + k
        t_upd, _ = event_time_ms(lambda: ops.mu_update_h(Hi, Si, Gi, p.eps, False), reps=20, warm=10)
        del Hi, Si
        fl_nt = 2.0 * m_l * n * k + 2.0 * m_l * k * k
        fl_tn = 2.0 * m_l * n * k
        fl_res = 2.0 * m_l * n * k + 3.0 * m_l * n
        by_upd = 12.0 * n_iso * k
        kt = max(1, (k + 31) // 32)
        small = k <= 16 and n % 32 == 0
        kname_nt = "nt16_kernel<FUSED_W>" if small else "nt_kernel<KT=%d,FUSED_W>" % kt
        kname_tn = "tn16_kernel" if (small and n % 64 == 0) else "tn_kernel<KT=%d,PARTIAL>" % kt

        def mfma_entry(kernel, ms_, flops, role):
            ach = flops / ms_ / 1e9
            e = {"kernel": kernel, "bound": "mfma", "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                 "frac": ach / PEAK_FP32_MFMA_TFLOPS, "traffic": None, "flops_per_launch": flops, "ms_per_launch": ms_}
            tr = pmc_traffic(role) if (m, n, k, world) == (262144, 8192, 64, 1) else None
            if tr is not None:
                e["traffic"] = tr["bytes"]
                e["traffic_note"] = ("HBM bytes per launch from the committed PMC pass %s, kernel %s (FETCH_SIZE x2 + "
                                     "WRITE_SIZE); a constant of that profile, not a live counter" % (tr["source"], tr["kernel"]))
            hc = out.get("held_clock_ghz") if out else None
            if hc:        # measured live over the whole step (dnmf_clock_probe): the peak above is at PEAK_CLOCK_GHZ
                e["held_clock_ghz_of_step"] = hc
                e["frac_at_held_clock"] = e["frac"] * PEAK_CLOCK_GHZ / hc
            return e

        def hbm_entry(kernel, ms_, nbytes, note, role=None, workload="bench"):
            ach = nbytes / ms_ / 1e6
            e = {"kernel": kernel, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                 "frac": ach / PEAK_HBM_GBS, "traffic": None, "bytes_per_launch": nbytes, "ms_per_launch": ms_,
                 "note": note}
            tr = pmc_traffic(role, workload) if (role and (m, n, k, world) == (262144, 8192, 64, 1)) else None
            if tr is not None:
                e["traffic"] = tr["bytes"]
                e["traffic_note"] = ("HBM bytes per launch from the committed PMC pass %s, kernel %s (FETCH_SIZE x2 + "
                                     "WRITE_SIZE); a constant of that profile, not a live counter" % (tr["source"], tr["kernel"]))
            return e

        kern = {
            "dnmf_aht_update_w = %s (A.H^T + W update, 1 launch)" % kname_nt: {
                "ms": t["aht_update_w"], "tflops": fl_nt / t["aht_update_w"] / 1e9,
                "frac_mfma": fl_nt / t["aht_update_w"] / 1e9 / PEAK_FP32_MFMA_TFLOPS,
                "algorithmic_gbs": (4.0 * m_l * n + 8.0 * m_l * k) / t["aht_update_w"] / 1e6},
            "dnmf_wta = %s + reduce_partials (W^T.A)" % kname_tn: {
                "ms": t["wta"], "tflops": fl_tn / t["wta"] / 1e9,
                "frac_mfma": fl_tn / t["wta"] / 1e9 / PEAK_FP32_MFMA_TFLOPS,
                "algorithmic_gbs": (4.0 * m_l * n + 4.0 * m_l * k) / t["wta"] / 1e6},
            "dnmf_mu_update_h = update_h_seq_kernel, in the step (H is %.1f MiB: latency bound)" % (4.0 * k * n / 2**20): {
                "ms": t["mu_update_h"], "algorithmic_gbs": 12.0 * n * k / t["mu_update_h"] / 1e6},
            "dnmf_mu_update_h = update_h_seq_kernel, isolated on %d x 2^22 (eltwise multiply-divide + k x k product)" % k: {
                "ms": t_upd, "algorithmic_gbs": by_upd / t_upd / 1e6, "frac_hbm": by_upd / t_upd / 1e6 / PEAK_HBM_GBS},
            "dnmf_gram_wtw (tn_kernel + reduce)": {"ms": t["gram_wtw"]},
            "dnmf_gram_hht (nt_kernel split + reduce)": {"ms": t["gram_hht"]},
            "dnmf_sqnorm = sqnorm_kernel (||A||^2, once per fit)": {
                "ms": t_sq, "algorithmic_gbs": 4.0 * m_l * n / t_sq / 1e6, "frac_hbm": 4.0 * m_l * n / t_sq / 1e6 / PEAK_HBM_GBS},
            "dnmf_resid_sqnorm = resid_lds_kernel (||A - W H||^2, once per fit)": {
                "ms": t_res, "tflops": fl_res / t_res / 1e9, "frac_mfma": fl_res / t_res / 1e9 / PEAK_FP32_MFMA_TFLOPS,
                "algorithmic_gbs": 4.0 * m_l * n / t_res / 1e6},
        }
        if rank == 0:
            r_nt = mfma_entry(kname_nt + " (dnmf_aht_update_w)", t["aht_update_w"], fl_nt, "nt")
            r_tn = mfma_entry(kname_tn + " (dnmf_wta, incl. the reduction of the partial slabs)", t["wta"], fl_tn, "tn")
            r_res = mfma_entry("resid_lds_kernel<KT=%d> (dnmf_resid_sqnorm)" % kt, t_res, fl_res, "resid_lds_kernel")
            r_upd = hbm_entry("update_h_seq_kernel<KT=%d> (dnmf_mu_update_h: H *= S / (G H + eps), %d x 2^22)" % (kt, k),
                              t_upd, by_upd, "algorithmic bytes = 12 per element of H (read H, read S, write H)",
                              role="update_h_seq_kernel<2", workload="elt")
            r_sq = hbm_entry("sqnorm_kernel (dnmf_sqnorm on X)", t_sq, 4.0 * m_l * n, "one read of X", role="sqnorm_kernel")
            out["roofline"] = r_nt
            out["roofline_hbm"] = r_upd
            out["rooflines"] = [r_nt, r_tn, r_res, r_upd, r_sq]
            if x6_t is not None:
                by_nt, by_tn = 4.0 * m_l * n + 8.0 * m_l * k, 4.0 * m_l * n + 4.0 * m_l * k + 4.0 * k * n
                out["rooflines"] += [
                    hbm_entry("ntx_kernel<KT=%d,FUSED_W> (dnmf_aht_update_w_bf16x6; incl. cutting H)" % kt, x6_t["aht_update_w"], by_nt,
                              "bf16x6 arithmetic; algorithmic bytes = X once + W read and written", role="ntx_kernel<2, 1", workload="split"),
                    hbm_entry("tnx_kernel<KT=%d> (dnmf_wta_bf16x6; incl. cutting W^T and the reduction of the partial slabs)" % kt, x6_t["wta"], by_tn,
                              "bf16x6 arithmetic; algorithmic bytes = X once + W once + the k x n result", role="tnx_kernel<2", workload="split")]
                if a.gemm == "bf16x6":
                    out["roofline"] = out["rooflines"][-2]
            out["kernels"] = kern

    if rank == 0:
        if not multi and not a.no_cpu_baseline:
            torch.cuda.synchronize()
            out["cpu_baseline"] = cpu_baseline(n, k, m)
            out["cpu_baseline_allcores"] = cpu_baseline_allcores(n, k, m)
        # LAST (it must not disturb the clocks of the measurements above).  Informational: the reference's OWN example next to its headline
        # configuration -- examples/dist_pynmfk_2d_Swim.py (NMFk, KL / MU, k = 14..18, 20 perturbations x 5000 iterations, 1024 x 256) on one rank, whole sweep incl. clustering and the regression fits; the known
        # answer is nopt == 16.  Small problems: the whole-fit kernels of csrc/dnmf_small.h (tools/swimbench.py is the stand-alone tool).
        swim_npz = os.path.join(ROOT, "tests", "golden", "data_swim.npz")
        if not multi and not a.no_swim and a.gemm == "fp32" and os.path.exists(swim_npz):
            try:
                out["reference_example_swim"] = swim_example(swim_npz, dev)
                if not a.no_cpu_baseline:
                    # the oracle's MU/KL step on a problem of the example's size, in the reference's process model (1 BLAS thread per process)
                    cores, model = host_cpu()
                    P = min(8, cores)
                    got = _cpu_pool(_cpu_kl_rank, [(r, 1024, 256, 16, 200) for r in range(P)], timeout=120)
                    if got:
                        sec = max(got.values())
                        steps = 5 * 21 * 5000
                        out["reference_example_swim"]["cpu_baseline"] = {
                            "value": P / sec, "unit": "KL steps/s", "cores": P, "kind": "port", "host_cpu": model, "seconds_per_step_per_core": sec,
                            "sweep_seconds_if_spread_over_the_cores": steps * sec / P,
                            "sample": "oracle kl_mu_step_local on a 1024 x 256, k = 16 problem: %d processes x 1 BLAS thread, 1 warm-up + 200 timed steps "
                                      "each (slowest process); the sweep is 525 000 such steps" % P}
            except Exception as ex:  # noqa: BLE001 -- informational: never costs the headline line
                out["reference_example_swim"] = {"error": str(ex)[:200]}
        if not multi and not a.no_configs and a.config == 3 and a.gemm == "fp32" and (m, n, k) == (262144, 8192, 64):
            torch.cuda.synchronize()
            out["configs"] = other_configs()
        flush_c_stdio()                                           # (RCCL's version banner sits in C stdio until here)
        print(json.dumps(out), flush=True)
    if world > 1:
        barrier()
    if multi:
        nc = getattr(p, "_native_comm", None)
        if nc is not None:
            nc.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
