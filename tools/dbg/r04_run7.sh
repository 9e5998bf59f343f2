cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
timeout 900 python -m pytest tests/test_bench_launch.py tests/test_gpu_nmfk.py -x -q -m gpu -k "config or two_rank or refuses" > gpurun_out/r04c/pytest.log 2>&1
tail -30 gpurun_out/r04c/pytest.log
for c in 2 5; do timeout 900 python bench.py --config $c > gpurun_out/r04c/config$c.json 2> gpurun_out/r04c/config$c.err; tail -c 1500 gpurun_out/r04c/config$c.json; done
timeout 900 python bench.py --config 4 --emulate-ranks 8 > gpurun_out/r04c/config4_emu8.json 2> gpurun_out/r04c/config4_emu8.err; tail -c 2500 gpurun_out/r04c/config4_emu8.json
timeout 900 python bench.py --config 4 --steps 10 > gpurun_out/r04c/config4_n1.json 2> gpurun_out/r04c/config4_n1.err; tail -c 2500 gpurun_out/r04c/config4_n1.json; tail -5 gpurun_out/r04c/config4_n1.err
