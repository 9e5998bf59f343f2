#!/bin/bash
# Run ON THE GPU BOX: one-pass vs two-pass MU/FRO step over a grid of shapes (shipped library).
R=${GRAFT_REPO_ROOT:-$(pwd)}
for shp in "$@"; do
  REPS=40 timeout 200 python3 $R/tools/teambench.py $shp 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print(d['m'], d['n'], d['k'], 'two', min(d['two_pass_ms_0'], d['two_pass_ms_1']), 'one', min(d['one_pass_ms_0'], d['one_pass_ms_1']), 'errW', '%.1e' % d['relerr_W_one'], 'errH', '%.1e' % d['relerr_H_one'], d['bit_identical_rerun'], d['timed_out'])"
done
