#!/bin/bash
# round 5: SQ counters of kl_uht_pipe_kernel at k = 32 with 32-row (DNMF_KLUHT_MR=0) and 64-row (2) wave tiles
R=${GRAFT_REPO_ROOT:-$(pwd)}
export DNMF_LIB_PATH=$R/tools/_build/libdnmf_hip_tune.so
export REPS=10
cd /tmp && export TMPDIR=/tmp
for k in 32 64; do
for v in 0 2 3; do
  export DNMF_KLUHT_MR=$v
  for set in "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
    tag=$(echo $set | cut -c4-12 | tr ' ' '_')
    OUT=$R/gpurun_out/mrpmc/k${k}_mr${v}_$tag
    mkdir -p $OUT
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT -- python3 $R/tools/uhtbench.py 32768 16384 $k > $OUT/stdout.txt 2> $OUT/log.txt
    f=$(find $OUT -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python3 - "$f" "k=$k MR=$v" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list); dur = []
for r in csv.DictReader(open(sys.argv[1])):
    if "kl_uht_pipe_kernel" not in r["Kernel_Name"]: continue
    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(sys.argv[2], "us=%.1f" % (sum(dur) / max(1, len(dur))), {k: round(sum(v) / len(v), 1) for k, v in agg.items()})
PY
    else echo "k=$k MR=$v [$set]: no counters ($(tail -1 $OUT/log.txt | cut -c1-120))"; fi
    rm -rf $OUT
  done
done
done
