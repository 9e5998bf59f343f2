cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h; mkdir -p $O
( time timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_output.json 2> $O/bench.err ) 2> $O/bench.time
timeout 900 python bench.py --config 2 > $O/config2_output.json 2> $O/config2.err
timeout 900 python bench.py --config 4 --emulate-ranks 8 > $O/config4_output.json 2> $O/config4.err
timeout 900 python bench.py --config 4 --steps 10 > $O/config4_n1_output.json 2> $O/config4_n1.err
timeout 900 python bench.py --config 5 > $O/config5_output.json 2> $O/config5.err
timeout 300 python bench.py --emulate-ranks 8 --steps 100 --no-kernel-timing > $O/emu8_output.json 2> $O/emu8.err
cat $O/bench.time
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04h/*_output.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], d["metric"], round(d["value"],2), d["unit"], "ms/step", round(d["ms_per_step"],3), "roofline", round(d["roofline"]["frac"],3) if d.get("roofline") else None, d.get("estimated_k"))
    except Exception as e:
        print(f, "FAILED", e)
PY
