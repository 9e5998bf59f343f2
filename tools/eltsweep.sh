#!/bin/bash
# Run ON THE GPU BOX: A/B sweep of the update-kernel variants of the TUNING build (tools/_build/libdnmf_hip_tune.so) on
# the isolation shape of SURVEY 8d (k x 2^22: 3.2 GB per pass at k = 64).  Usage: tools/eltsweep.sh [k ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
export DNMF_LIB_PATH=$R/tools/_build/libdnmf_hip_tune.so
for k in ${@:-32 64 128}; do
  for h in 0 14 15 16 18 23 24 25; do
    for w in 4; do
      echo "k=$k DNMF_UPD_H=$h DNMF_UPD_W=$w $(DNMF_UPD_H=$h DNMF_UPD_W=$w ELT=mu_update_h python3 $R/tools/eltbench.py $k 2>&1 | tail -1)"
    done
  done
  for w in 0 2 3 4 5 6 8; do
    echo "k=$k DNMF_UPD_W=$w $(DNMF_UPD_H=14 DNMF_UPD_W=$w ELT=mu_update_w python3 $R/tools/eltbench.py $k 2>&1 | tail -1)"
  done
done
