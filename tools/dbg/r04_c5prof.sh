cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r04k
rm -rf /tmp/c5p; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5p -- python3 $R/bench.py --config 5 --itr 10 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/r04k/c5_itr10.json 2> /dev/null
f=$(find /tmp/c5p -name "*kernel_stats.csv" | head -1)
head -25 $f | cut -d, -f1-4 | cut -c1-150
python3 - <<PY
import json
d=json.loads([l for l in open("$R/gpurun_out/r04k/c5_itr10.json") if l.startswith("{")][-1])
print("sweep seconds", d["seconds_per_sweep"])
import csv
tot=0
for r in csv.DictReader(open("$f")):
    tot+=float(r["TotalDurationNs"])
print("sum of kernel time (whole process, warm-up included) s:", tot/1e9)
PY
cd $R && python3 - <<'PY'
import cProfile, pstats, sys, io, contextlib
sys.argv=["bench.py","--config","5","--itr","10","--no-cpu-baseline","--no-kernel-timing"]
import bench
pr=cProfile.Profile(); pr.enable()
with contextlib.redirect_stdout(io.StringIO()):
    try: bench.main()
    except SystemExit: pass
pr.disable()
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:6000])
PY
