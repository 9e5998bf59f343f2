cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/c5p; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5p -- python3 $R/bench.py --config 5 --itr 100 --perturbations 4 --no-cpu-baseline --no-kernel-timing > /tmp/c5.json 2> /dev/null
f=$(find /tmp/c5p -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv, json
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:60]
    print("%-62s calls %6s avg %8.1f us total %7.1f ms %5.1f%%"%(n, r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, 100*float(r["TotalDurationNs"])/tot))
d=json.loads([l for l in open("/tmp/c5.json") if l.startswith("{")][-1])
print("sweep s", d["seconds_per_sweep"], "kernel total s (incl warmup)", tot/1e9)
PY
