cd $GRAFT_REPO_ROOT
export DNMF_LIB_PATH=tools/_build/libdnmf_hip_tune.so
mkdir -p gpurun_out/r04a; L=gpurun_out/r04a/coal.log; : > $L
run() { timeout 120 python tools/uhtbench.py $1 $2 $3 2>/dev/null | grep which >> $L; }
for a in 0 64; do DNMF_KLUHT_VAR=410 DNMF_KLUHT_ABL=$a run 32768 16384 32; done
DNMF_KLUHT_VAR=411 run 32768 16384 32
for a in 0 64; do DNMF_KLUHT_VAR=310 DNMF_KLUHT_ABL=$a run 32768 16384 64; done
KB=wtu run 32768 16384 32
KB=wtu run 32768 16384 64
cat $L
