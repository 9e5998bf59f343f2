#!/bin/bash
# round 5: update_w16_kernel (16-row wave tiles) against update_w_seq_kernel on the isolation pass; needs the tuning build
export DNMF_LIB_PATH=tools/_build/libdnmf_hip_tune.so
DNMF_UPD_W16=1 python -m pytest tests/test_gpu_kernels.py -q -k "mu_updates" 2>&1 | tail -3
for k in 32 64 128; do
  for v in 0 1 1024 2048 4096; do
    echo -n "k=$k W16=$v: "; DNMF_UPD_W16=$v ELT=mu_update_w python tools/eltbench.py $k 2>&1 | tail -1
  done
done
