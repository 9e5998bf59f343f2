for m in 32768 65536; do
  KB=nt,tn python tools/kbench.py $m 8192 64
done
python bench.py --rows 32768 --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step']); print(json.dumps(d.get('kernels'),indent=0))"
