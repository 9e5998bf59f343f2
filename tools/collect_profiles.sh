#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for the bench workload.
# Usage: tools/collect_profiles.sh <tag>   -> writes gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline"
SHORT="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $SHORT > /dev/null 2> $OUT/pmc_sq.log
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $SHORT > /dev/null 2> $OUT/pmc_fetch.log
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- $SHORT > /dev/null 2> $OUT/pmc_write.log
cd $R
python3 tools/summarize_profiles.py $OUT $TAG
