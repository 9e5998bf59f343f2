cd $GRAFT_REPO_ROOT
export DNMF_LIB_PATH=tools/_build/libdnmf_hip_tune.so
mkdir -p gpurun_out/r04a; L=gpurun_out/r04a/wtu.log; : > $L
run() { KB=wtu timeout 120 python tools/uhtbench.py $1 $2 $3 2>/dev/null | grep which | sed "s/}/, \"wvar\": \"$DNMF_WTU_VAR nt1=$DNMF_WTU_NT1\"}/" >> $L; }
export DNMF_WTU_NT1=4
for v in 0; do DNMF_WTU_VAR=$v run 32768 16384 32; DNMF_WTU_VAR=$v run 65536 4096 32; DNMF_WTU_VAR=$v run 32768 16384 24; done
export DNMF_WTU_NT1=2
for v in 22 23; do DNMF_WTU_VAR=$v run 32768 16384 32;  DNMF_WTU_VAR=$v run 65536 4096 32; DNMF_WTU_VAR=$v run 32768 16384 24; done
export DNMF_WTU_NT1=4
for v in 0 101; do DNMF_WTU_VAR=$v run 32768 16384 64; done
cat $L
