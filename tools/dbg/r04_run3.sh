cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
export DNMF_LIB_PATH=$R/tools/_build/libdnmf_hip_tune.so
export REPS=10
mkdir -p $R/gpurun_out/r04a
cd /tmp && export TMPDIR=/tmp
for a in 0 4 1 2 8; do
  export DNMF_KLUHT_VAR=410 DNMF_KLUHT_ABL=$a
  [ $a = 0 ] && export DNMF_KLUHT_VAR=0
  rm -rf /tmp/pm; timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm -- python3 $R/tools/uhtbench.py 32768 16384 32 > /dev/null 2>&1
  echo "abl=$a" >> $R/gpurun_out/r04a/pmc_abl.log
  python3 $R/tools/pmc_summary.py /tmp/pm kl_uht >> $R/gpurun_out/r04a/pmc_abl.log
  rm -rf /tmp/pm; timeout 300 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM_RD TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum --output-format csv -d /tmp/pm -- python3 $R/tools/uhtbench.py 32768 16384 32 > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pm kl_uht >> $R/gpurun_out/r04a/pmc_abl.log
done
cat $R/gpurun_out/r04a/pmc_abl.log
