#!/bin/bash
# Run ON THE GPU BOX: A/B of the one-pass team kernel's tuning switches (tuning build), one JSON line per variant.
# usage: tools/team_ab.sh "SD NT X" ...      (X: 1 = take granules as they are, 2 = no exchange, 4 = plain stores)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export DNMF_LIB_PATH=$R/tools/_build/libdnmf_hip_tune.so
for v in "$@"; do
  set -- $v
  echo "SD=$1 NT=$2 X=$3"
  DNMF_TEAM_SD=$1 DNMF_TEAM_NT=$2 DNMF_TEAM_X=$3 REPS=40 NOTWO=1 timeout 100 python3 $R/tools/teambench.py 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d[k] for k in d if 'ms' in k or k in ('relerr_W_one', 'bit_identical_rerun', 'timed_out')})"
done
