R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export KB=nt
for mode in real alias; do
  if [ $mode = alias ]; then export ALIAS=1; else unset ALIAS; fi
  O=$R/gpurun_out/pmc_lat_$mode
  rm -rf $O; mkdir -p $O
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD --output-format csv -d $O -- python3 $R/tools/kbench.py 65536 8192 64 > /dev/null 2> $O/log.txt
  echo "== $mode"; python3 $R/tools/pmc_summary.py $O nt_kernel
done
