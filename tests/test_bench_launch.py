"""bench.py as the driver (and a person) launches it: `python bench.py --gpus N` starts its own N ranks, N > 1 lines carry
the exchange diagnostics.  CPU tier: the launcher refuses cleanly without GPUs.  GPU tier: two ranks stacked on the one
GPU over the host-staged gloo transport (RCCL refuses two ranks on one device) -- the whole N > 1 code path of bench.py
except the RCCL calls themselves, which tests/test_gpu_rccl.py covers on a one-rank group."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_count_gpus_needs_no_hip():
    sys.path.insert(0, ROOT)
    import bench
    n = bench.count_gpus()
    assert isinstance(n, int) and n >= 0


@pytest.mark.skipif(os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK), reason="a GPU is present")
def test_launcher_refuses_without_gpus():
    """No GPU in this container: `--gpus 2` must end with ONE line saying why and a non-zero exit, before any rank starts."""
    r = _run(["--gpus", "2", "--steps", "2"], timeout=120)
    assert r.returncode == 2, (r.returncode, r.stderr)
    err = [ln for ln in r.stderr.splitlines() if ln.strip()]
    assert len(err) == 1 and "--gpus 2" in err[0] and "GPU" in err[0], r.stderr
    assert r.stdout.strip() == ""


@pytest.mark.gpu
@pytest.mark.parametrize("cols", [1024, 4096])
def test_two_ranks_on_one_gpu_gloo(cols):
    r = _run(["--gpus", "2", "--rows", "8192", "--cols", str(cols), "--steps", "5", "--warmup", "2", "--backend", "gloo",
              "--no-cpu-baseline"], env_extra={"DNMF_BENCH_OVERSUBSCRIBE": "1"})
    assert r.returncode == 0, r.stderr[-4000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 2
    assert out["scaling"] == "strong" and out["value"] > 0 and out["config"]["rows_per_gpu"] == 4096
    mg = out["multi_gpu"]
    assert mg["rccl_ranks_seen"] == 2 and mg["backend"] == "gloo"
    assert len(mg["compute_only_ms_per_rank"]) == 2 and all(v > 0 for v in mg["compute_only_ms_per_rank"])
    assert abs(mg["exposed_comm_ms"] - (mg["full_step_ms"] - mg["compute_only_ms"])) < 1e-9
    assert set(mg["exchange_ab_ms_per_step"]) == {"torch/chunks=1", "torch/chunks=2", "torch/chunks=4"}
    assert mg["overlap_chunks_used"] in (1, 2, 4) and mg["exchange_used"] == "torch"
    assert mg["allreduce_alone_ms"] > 0 and mg["allreduce_bytes"] == 4 * (64 * cols + 64 * 64)
    assert out["roofline"]["bound"] == "mfma" and "cpu_baseline" not in out


@pytest.mark.gpu
def test_launcher_refuses_more_ranks_than_gpus_over_rccl():
    """One GPU on the box: `--gpus 2` over RCCL must be refused by the parent (one line, exit 2), not fail inside RCCL."""
    sys.path.insert(0, ROOT)
    import bench
    ng = bench.count_gpus() + 1            # one more than the node has (the box of the GPU tier has one; any node works)
    r = _run(["--gpus", str(ng), "--steps", "2"], timeout=120)
    assert r.returncode == 2 and ("--gpus %d" % ng) in r.stderr, (r.returncode, r.stderr)


@pytest.mark.gpu
def test_single_gpu_line_small():
    r = _run(["--rows", "8192", "--cols", "1024", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-sustained"])
    assert r.returncode == 0, r.stderr[-4000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 1 and "multi_gpu" not in out and out["value"] > 0


@pytest.mark.gpu
def test_emulated_rank_runs_the_rccl_exchange_paths():
    """`--emulate-ranks 8` on the one GPU: one rank's share of the 8-GPU row grid with the REAL RCCL calls on a one-rank
    group -- both exchange transports (torch.distributed and the library's own communicator) x 1 / 2 / 4 overlap chunks are
    timed in the warm-up, i.e. every line of bench.py's N > 1 path runs over RCCL before a multi-GPU node ever sees it."""
    r = _run(["--emulate-ranks", "8", "--rows", "65536", "--cols", "4096", "--steps", "10", "--warmup", "2", "--no-kernel-timing"])
    assert r.returncode == 0, r.stderr[-4000:]
    out = _json_line(r.stdout)
    mg = out["multi_gpu"]
    assert out["n_gpus"] == 1 and "EMULATED" in out["config"]["workload"] and out["config"]["rows_per_gpu"] == 8192
    assert mg["backend"] == "nccl" and "native_exchange_unavailable" not in mg
    assert set(mg["exchange_ab_ms_per_step"]) == {"%s/chunks=%d" % (t, c) for t in ("torch", "native") for c in (1, 2, 4)}
    assert mg["exchange_used"] in ("torch", "native") and mg["overlap_chunks_used"] in (1, 2, 4)
    assert mg["compute_only_ms"] > 0 and mg["allreduce_alone_ms"] > 0


# ---- round 4: every BASELINE configuration is launchable as typed
@pytest.mark.gpu
def test_config4_emulated_rank_of_the_4x2_grid():
    """`--config 4 --emulate-ranks 8` on the one GPU: rank 0 of the 4 x 2 grid steps its own block through nmf_algorithms_2D
    with the real collective calls on one-rank groups; the line names the configuration and carries the KL roofline."""
    r = _run(["--config", "4", "--emulate-ranks", "8", "--rows", "8192", "--cols", "4096", "--steps", "4", "--warmup", "1",
              "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-4000:]
    out = _json_line(r.stdout)
    assert out["metric"] == "mu_iterations_per_sec" and out["n_gpus"] == 1 and out["value"] > 0
    assert "BASELINE config 4" in out["config"]["workload"] and "EMULATED" in out["config"]["workload"]
    assert out["config"]["block_per_gpu"] == [2048, 2048] and out["config"]["k"] == 128
    mg = out["multi_gpu"]
    # both arms of the exchange A/B ran: torch.distributed between the launches, and the whole step inside the library
    # (dnmf_mu_kl_step_2d over dnmf_comm_create_emulated); the faster one is used for the timed region
    assert mg["grid"] == [4, 2] and mg["rccl_ranks_seen"] == 1 and mg["exchange_used"] in ("torch", "native")
    assert set(mg["exchange_ab_ms_per_step"]) == {"torch", "native"} and all(v > 0 for v in mg["exchange_ab_ms_per_step"].values())
    assert mg["compute_only_ms"] > 0 and abs(mg["exposed_comm_ms"] - (mg["full_step_ms"] - mg["compute_only_ms"])) < 1e-9
    rf = out["roofline"]
    assert rf["bound"] == "mfma" and rf["kernel"].startswith("kl_uht_pipe_kernel<KT=4>") and 0 < rf["frac"] < 1
    assert rf["flops_per_launch"] == 4.0 * 2048 * 2048 * 128 and "cpu_baseline" not in out


@pytest.mark.gpu
def test_config4_on_a_2x2_grid_stacked_on_one_gpu():
    """`--config 4 --gpus 4` over gloo with the four ranks stacked on the one GPU: the whole N > 1 path of config 4 (2 x 2
    grid, sub-communicators, allgather / reduce-scatter between the kernels, compute-only leg)."""
    r = _run(["--config", "4", "--gpus", "4", "--backend", "gloo", "--rows", "2048", "--cols", "1024", "--rank", "32", "--steps", "3",
              "--warmup", "1", "--no-kernel-timing"], env_extra={"DNMF_BENCH_OVERSUBSCRIBE": "1"})
    assert r.returncode == 0, r.stderr[-4000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 4 and out["scaling"] == "strong" and out["config"]["block_per_gpu"] == [1024, 512]
    mg = out["multi_gpu"]
    assert mg["grid"] == [2, 2] and mg["rccl_ranks_seen"] == 4 and len(mg["compute_only_ms_per_rank"]) == 4
    assert "2D grid p_r=2 p_c=2" in out["config"]["workload"]


@pytest.mark.gpu
def test_config5_sweep_line_small():
    """`--config 5` cut down (k = 2..5, 3 perturbations x 20 HALS iterations on 4096 x 512): one sweep = one step, fits/s."""
    r = _run(["--config", "5", "--rows", "4096", "--cols", "512", "--end-k", "5", "--perturbations", "3", "--itr", "20",
              "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-4000:]
    out = _json_line(r.stdout)
    assert out["metric"] == "nmfk_fits_per_sec" and out["unit"] == "fits/s" and out["fits_per_step"] == 12 and out["value"] > 0
    assert "BASELINE config 5" in out["config"]["workload"] and out["dtype"].startswith("bf16")
    assert out["roofline"]["bound"] == "hbm" and out["roofline"]["kernel"].startswith("nt16_kernel<bf16 X>")


@pytest.mark.gpu
def test_config5_two_ranks_share_the_perturbations():
    """`--config 5 --gpus 2` (gloo, both ranks on the one GPU): the ranks hold the whole X and share the perturbation fits
    (params.nmfk_split = 'perturbations'); the planted rank must come back, and the single-rank sweep must agree on it."""
    common = ["--config", "5", "--rows", "4096", "--cols", "512", "--end-k", "8", "--perturbations", "4", "--itr", "60",
              "--no-cpu-baseline", "--no-kernel-timing"]
    r = _run(common + ["--gpus", "2", "--backend", "gloo"], env_extra={"DNMF_BENCH_OVERSUBSCRIBE": "1"})
    assert r.returncode == 0, r.stderr[-4000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["fits_per_step"] == 28 and out["value"] > 0
    assert out["config"]["parallelism"].startswith("perturbations over 2 ranks") and out["config"]["block_per_gpu"] == [4096, 512]
    one = _json_line(_run(common).stdout)
    assert out["estimated_k"] == one["estimated_k"] == 6


@pytest.mark.gpu
def test_config2_line_small_steps():
    r = _run(["--config", "2", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-sustained"])
    assert r.returncode == 0, r.stderr[-4000:]
    out = _json_line(r.stdout)
    assert "BASELINE config 2" in out["config"]["workload"] and (out["config"]["m"], out["config"]["n"], out["config"]["k"]) == (65536, 4096, 32)
    assert out["roofline"]["bound"] == "mfma" and out["value"] > 500


@pytest.mark.gpu
def test_three_exchange_arms_over_gloo_with_a_hosted_library_communicator():
    """The warm-up A/B of an N > 1 run with all THREE arms -- torch.distributed between the launches, the library-sequenced step, and
    the library-sequenced step with its packed allreduce over IPC peer buffers (`native-direct`) -- on two ranks stacked on the
    one GPU: the library's communicator is a hosted one over gloo here (DNMF_BENCH_HOSTED=1; on a node it is RCCL)."""
    r = _run(["--gpus", "2", "--backend", "gloo", "--rows", "8192", "--cols", "1024", "--steps", "6", "--warmup", "2", "--no-kernel-timing",
              "--no-sustained"], env_extra={"DNMF_BENCH_OVERSUBSCRIBE": "1", "DNMF_BENCH_HOSTED": "1"})
    assert r.returncode == 0, r.stderr[-4000:]
    mg = _json_line(r.stdout)["multi_gpu"]
    assert set(mg["exchange_ab_ms_per_step"]) == {"%s/chunks=%d" % (t, c) for t in ("torch", "native", "native-direct") for c in (1, 2, 4)}
    assert mg["direct_vs_rccl_max_rel_diff"] < 1e-5 and mg["exchange_used"] in ("torch", "native", "native-direct")
    assert mg["native_transport"] == "hosted over gloo"
