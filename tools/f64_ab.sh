#!/bin/bash
# Run ON THE GPU BOX: A/B of the float64 kernels' tuning switches (tuning build), one line per variant and kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}
export DNMF_LIB_PATH=$R/tools/_build/libdnmf_hip_tune.so
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', ' | '.join('%s %.3f ms %.3f' % (k.split(' ')[0], v['ms'], v['frac_fp64_mfma']) for k, v in d['kernels'].items()), '| fro %.3f kl %.3f' % (d['mu_fro_step']['ms'], d['mu_kl_step']['ms']))"; }
python3 $R/tools/f64bench.py "$@" 2>/dev/null | show base
DNMF_F64_IL=0 python3 $R/tools/f64bench.py "$@" 2>/dev/null | show no_il
DNMF_F64_NT_D=3 DNMF_F64_TN_D=3 DNMF_F64_KL_D=3 python3 $R/tools/f64bench.py "$@" 2>/dev/null | show depth3
