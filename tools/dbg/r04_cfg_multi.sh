cd $GRAFT_REPO_ROOT
export DNMF_BENCH_OVERSUBSCRIBE=1
for args in "--config 4 --gpus 2 --backend gloo --rows 2048 --cols 1024 --rank 32 --steps 3 --warmup 1 --no-kernel-timing" \
            "--config 5 --gpus 2 --backend gloo --rows 4096 --cols 512 --end-k 4 --perturbations 2 --itr 10 --no-kernel-timing" \
            "--config 5 --gpus 2 --backend gloo --grid 2x1 --rows 4096 --cols 512 --end-k 4 --perturbations 2 --itr 10 --no-kernel-timing" \
            "--config 2 --gpus 2 --backend gloo --steps 5 --warmup 1 --no-kernel-timing"; do
  echo "== $args"
  timeout 300 python bench.py $args 2> /tmp/err.log | tail -c 900; echo; tail -3 /tmp/err.log | cut -c1-300
done
