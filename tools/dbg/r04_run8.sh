cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "kl" > gpurun_out/r04c/pytest2.log 2>&1; tail -3 gpurun_out/r04c/pytest2.log
timeout 900 python bench.py --config 5 > gpurun_out/r04c/config5.json 2> gpurun_out/r04c/config5.err
timeout 900 python bench.py --config 4 --steps 10 --no-cpu-baseline > gpurun_out/r04c/config4_n1.json 2> gpurun_out/r04c/config4_n1.err; tail -3 gpurun_out/r04c/config4_n1.err
python - <<'PY'
import json
for f in ("config5","config4_n1"):
    d=json.loads([l for l in open('gpurun_out/r04c/%s.json'%f) if l.startswith('{')][-1])
    print(f, {k:d[k] for k in ("metric","value","unit","ms_per_step","steps")}, d.get("estimated_k"), d.get("hals_iterations_per_sec"), [ (r["kernel"][:20], round(r["frac"],3)) for r in d["rooflines"]])
PY
