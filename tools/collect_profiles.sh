#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes, condensed into profiles/<tag>_<workload>_*.
# Usage: tools/collect_profiles.sh <tag> [workload ...]      workloads: bench kl kl16 kl32 kl64 c4 c5 c5s c5s1 f64 swim wide elt128 hals16 hals64 k16 bf16 elt c2 team split klsplit klsplit128  (default: the first eight)
# The profiled program always stands directly after `--` (no env / bash -c / wrapper: the profiler's preloaded library has
# already initialised the GPU, a re-exec from there takes the box down).  PMC counters are collected in their own runs.
set -u
TAG=${1:-r02}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
WL=${@:-bench kl hals16 hals64 k16 bf16 elt c2}
cd /tmp && export TMPDIR=/tmp
prog() {   # the command line of a workload: "full" (stats pass) or "short" (PMC passes)
  case $1 in
    bench)  if [ "$2" = full ]; then echo "$R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sustained --no-configs"; else echo "$R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sustained --no-configs"; fi ;;
    kl)     echo "$R/tools/klbench.py 32768 32768 128" ;;
    kl16)   echo "$R/tools/klbench.py 32768 16384 16" ;;
    kl32)   echo "$R/tools/klbench.py 32768 16384 32" ;;
    kl64)   echo "$R/tools/klbench.py 32768 16384 64" ;;
    hals16) echo "$R/tools/halsbench.py 262144 8192 16" ;;
    hals64) echo "$R/tools/halsbench.py 262144 8192 64" ;;
    k16)    echo "$R/tools/kbench.py 262144 8192 16" ;;
    bf16)   echo "$R/tools/kbench.py 262144 8192 64 --bf16" ;;
    elt)    echo "$R/tools/eltbench.py 64" ;;
    elt128) echo "$R/tools/eltbench.py 128" ;;
    c4)     echo "$R/bench.py --config 4 --emulate-ranks 8 --steps 10 --warmup 2 --no-cpu-baseline" ;;
    c5)     echo "$R/bench.py --config 5 --no-cpu-baseline" ;;
    c5s)    echo "$R/bench.py --config 5 --rows 1024 --cols 256 --no-cpu-baseline --no-kernel-timing" ;;
    c5s1)   echo "$R/bench.py --config 5 --rows 1024 --cols 256 --no-cpu-baseline --no-kernel-timing --nmfk-batch 1" ;;
    f64)    echo "$R/tools/f64bench.py 65536 4096 64" ;;
    swim)   if [ "$2" = full ]; then echo "$R/tools/swimbench.py --reps 1"; else echo "$R/tools/swimbench.py --reps 1 --itr 300"; fi ;;
    wide)   echo "$R/tools/kbench.py 65536 4096 192" ;;
    c2)     echo "$R/tools/config_bench.py c2" ;;
    team)   echo "$R/tools/teambench.py" ;;
    team16) echo "$R/tools/teambench.py 65536 4096 16" ;;
    split)  echo "$R/tools/splitbench.py 262144 8192 64 --nocheck" ;;
    klsplit) echo "$R/tools/klsplitbench.py 32768 16384 16 --nocheck" ;;
    klsplit128) echo "$R/tools/klsplitbench.py 32768 32768 128 --nocheck" ;;
  esac
}
for w in $WL; do
  OUT=$R/gpurun_out/prof_${TAG}_$w
  rm -rf $OUT; mkdir -p $OUT
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $(prog $w full) > $OUT/stdout.json 2> $OUT/stats.log
  case $w in team|team16|bench|kl|kl16|kl32|kl64|elt|elt128|c4|hals64|c2|split|klsplit|klsplit128|f64|wide|swim)
    timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $(prog $w short) > /dev/null 2> $OUT/pmc_sq.log
    timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $(prog $w short) > /dev/null 2> $OUT/pmc_fetch.log
    timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $(prog $w short) > /dev/null 2> $OUT/pmc_write.log ;;
  esac
  (cd $R && python3 tools/summarize_profiles.py $OUT ${TAG}_$w "$(prog $w full | sed "s#$R/##")")
  # only gpurun_out/ travels back (64 MiB at most): the condensed artefacts and the program's own output go to
  # gpurun_out/profiles_out/ (copy them into profiles/), the raw traces are dropped on the box
  mkdir -p $R/gpurun_out/profiles_out
  cp $R/profiles/${TAG}_${w}_* $R/gpurun_out/profiles_out/ 2>/dev/null
  cp $OUT/stdout.json $R/gpurun_out/profiles_out/${TAG}_${w}_output.json 2>/dev/null
  rm -rf $OUT/stats $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write
done
