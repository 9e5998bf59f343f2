set -x
timeout 900 python -m pytest tests/test_gpu_bf16.py -x -q 2>&1 | tail -15
for k in 16 32 64; do
  KB=nt,tn,norm python tools/kbench.py 262144 8192 $k
  BF16=1 KB=nt,tn,norm python tools/kbench.py 262144 8192 $k
  BF16=1 DNMF_NT_PF=1 KB=nt python tools/kbench.py 262144 8192 $k
done
