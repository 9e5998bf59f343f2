cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04d; L=gpurun_out/r04d/updgrid2.log; : > $L
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r04d/pytest3.log 2>&1; tail -3 gpurun_out/r04d/pytest3.log
export DNMF_LIB_PATH=tools/_build/libdnmf_hip_tune.so
for g in 256 512 768 1024 2048 4096; do
  for e in mu_update_h mu_update_w; do echo "grid=$g" >> $L; DNMF_UPD_GRID=$g ELT=$e timeout 100 python tools/eltbench.py 128 2>/dev/null | grep bytes >> $L; done
done
cat $L
