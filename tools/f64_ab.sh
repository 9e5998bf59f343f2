#!/bin/bash
# Run ON THE GPU BOX: the float64 kernels at three ranks (one line per rank: kernel times, fraction of the fp64 MFMA peak, step times),
# then the switches the tuning build still has (row tiles per wave of the NT kernel / of the fused KL W-phase kernel).
# The variants that were measured and removed (three register sets, skewed column starts, four waves sharing a row block) are
# recorded in profiles/r06_f64_ab.txt.
R=${GRAFT_REPO_ROOT:-/root/repo}
export DNMF_LIB_PATH=$R/tools/_build/libdnmf_hip_tune.so
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['shape'], ' | '.join('%s %.3f ms %.3f' % (k.split(' ')[0], v['ms'], v['frac_fp64_mfma']) for k, v in d['kernels'].items()), '| fro %.3f kl %.3f' % (d['mu_fro_step']['ms'], d['mu_kl_step']['ms']))"; }
for k in 16 32 64; do python3 $R/tools/f64bench.py 65536 4096 $k 2>/dev/null | show base; done
for rt in 2 1; do DNMF_F64_NT_RT=$rt python3 $R/tools/f64bench.py 65536 4096 64 2>/dev/null | show nt_rt$rt; done
DNMF_F64_KL_RT=1 python3 $R/tools/f64bench.py 65536 4096 64 2>/dev/null | show kl_rt1
