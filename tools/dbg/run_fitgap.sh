cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mn in "mu kl" "mu fro" "hals fro"; do
  set -- $mn
  rm -rf /tmp/fg; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fg -- python3 $R/tools/dbg/fitgap.py $1 $2 > /tmp/fg_out.txt 2>/dev/null
  echo "== $1 $2"; cat /tmp/fg_out.txt | tail -1
  f=$(find /tmp/fg -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print("%-70s calls %6s avg %8.1f us total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
print("total kernel ms", tot/1e6)
PY
done
