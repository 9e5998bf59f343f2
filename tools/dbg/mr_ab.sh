#!/bin/bash
# round 5: 64-row wave tiles of kl_uht_pipe_kernel (DNMF_KLUHT_MR = 2 / 3) against the 32-row kernel: bit identity, then timings
export DNMF_LIB_PATH=tools/_build/libdnmf_hip_tune.so
mkdir -p /tmp/mrab
DNMF_KLUHT_MR=0 python tools/kluht_ab.py run mr0 /tmp/mrab > /dev/null
for v in 2 3; do DNMF_KLUHT_MR=$v python tools/kluht_ab.py run mr$v /tmp/mrab > /dev/null; python tools/kluht_ab.py compare mr0 mr$v /tmp/mrab | grep -c "bit identical"; done
for k in 32 64; do for v in 0 2 3; do echo -n "k=$k MR=$v: "; DNMF_KLUHT_MR=$v KB=uht python tools/uhtbench.py 32768 16384 $k | tail -1; done; done
