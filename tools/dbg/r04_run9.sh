cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04d
for k in 128 64 32; do timeout 200 python tools/eltbench.py $k >> gpurun_out/r04d/elt.log 2>&1; done
timeout 200 python tools/klbench.py 32768 16384 16 >> gpurun_out/r04d/kl16.log 2>&1
timeout 200 python tools/klbench.py 65536 4096 9 >> gpurun_out/r04d/kl16.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_bf16.py tests/test_capi.py -x -q > gpurun_out/r04d/pytest.log 2>&1
tail -4 gpurun_out/r04d/pytest.log
grep -v amdgpu.ids gpurun_out/r04d/elt.log gpurun_out/r04d/kl16.log
