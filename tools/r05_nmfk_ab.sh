#!/bin/bash
# round 5: the NMFk sweep before / after the whole-fit + batched entry points (bench.py --config 5), small (the reference's swim
# shape) and default (65536 x 4096) -- writes gpurun_out/r05_c5_*.json
mkdir -p gpurun_out
for shape in "--rows 1024 --cols 256" ""; do
  tag=$([ -n "$shape" ] && echo small || echo default)
  python bench.py --config 5 $shape --no-cpu-baseline --no-kernel-timing --fit-loop python --nmfk-batch 1 > gpurun_out/r05_c5_${tag}_before.json 2>gpurun_out/r05_c5_${tag}_before.err
  python bench.py --config 5 $shape --no-cpu-baseline --no-kernel-timing --nmfk-batch 1 > gpurun_out/r05_c5_${tag}_wholefit.json 2>gpurun_out/r05_c5_${tag}_wholefit.err
  python bench.py --config 5 $shape --no-cpu-baseline --no-kernel-timing > gpurun_out/r05_c5_${tag}_batched.json 2>gpurun_out/r05_c5_${tag}_batched.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_c5_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "fits/s %.2f" % d["value"], "s/sweep %.3f" % d["seconds_per_sweep"], "k_est", d["estimated_k"])
    except Exception as ex:
        print(f, "FAILED", ex)
PY
