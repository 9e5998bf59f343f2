cd $GRAFT_REPO_ROOT
export DNMF_LIB_PATH=tools/_build/libdnmf_hip_tune.so
mkdir -p gpurun_out/r04a; L=gpurun_out/r04a/abl2.log; : > $L
run() { timeout 120 python tools/uhtbench.py $1 $2 $3 2>>$L.err | grep which >> $L; }
for a in 0 128 136 129; do DNMF_KLUHT_VAR=410 DNMF_KLUHT_ABL=$a run 32768 16384 32; done
for a in 0 128; do DNMF_KLUHT_VAR=310 DNMF_KLUHT_ABL=$a run 32768 16384 64; done
cat $L; tail -5 $L.err
