set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
export T=tools/_build/libdnmf_hip_tune.so
DNMF_LIB_PATH=$T DNMF_KLUHT_PIPE=1 timeout 300 python tools/kluht_ab.py run pipe gpurun_out/r04a > gpurun_out/r04a/ab_pipe.log 2>&1
DNMF_LIB_PATH=$T DNMF_KLUHT_PIPE=0 timeout 300 python tools/kluht_ab.py run old gpurun_out/r04a > gpurun_out/r04a/ab_old.log 2>&1
timeout 100 python tools/kluht_ab.py compare pipe old gpurun_out/r04a > gpurun_out/r04a/ab_cmp.log 2>&1
rm -f gpurun_out/r04a/*.pt
for shape in "32768 32768 128" "32768 16384 64" "32768 16384 32" "65536 4096 32" "32768 16384 100"; do
  for p in 1 0; do
    echo "== $shape pipe=$p" >> gpurun_out/r04a/klbench.log
    DNMF_LIB_PATH=$T DNMF_KLUHT_PIPE=$p timeout 300 python tools/klbench.py $shape >> gpurun_out/r04a/klbench.log 2>&1
  done
done
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -m gpu -k "kl or KL" > gpurun_out/r04a/pytest_kl.log 2>&1
tail -5 gpurun_out/r04a/pytest_kl.log
cat gpurun_out/r04a/ab_cmp.log gpurun_out/r04a/klbench.log
