cd $GRAFT_REPO_ROOT
export DNMF_LIB_PATH=tools/_build/libdnmf_hip_tune.so
mkdir -p gpurun_out/r04a; L=gpurun_out/r04a/aux.log; : > $L
run() { timeout 120 python tools/uhtbench.py $1 $2 $3 2>/dev/null | grep which >> $L; }
for v in 0 411 412; do DNMF_KLUHT_VAR=$v run 32768 16384 32; done
for v in 0 311 312; do DNMF_KLUHT_VAR=$v run 32768 16384 64; done
for v in 0 201 202; do DNMF_KLUHT_VAR=$v run 32768 32768 128; done
for v in 0 411 412; do DNMF_KLUHT_VAR=$v run 65536 4096 32; done
cat $L
