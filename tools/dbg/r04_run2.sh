cd $GRAFT_REPO_ROOT
export DNMF_LIB_PATH=tools/_build/libdnmf_hip_tune.so
mkdir -p gpurun_out/r04a; L=gpurun_out/r04a/abl.log; : > $L
run() { timeout 120 python tools/uhtbench.py $1 $2 $3 2>/dev/null | grep which >> $L; }
for v in 0 300 400 200; do DNMF_KLUHT_VAR=$v run 32768 16384 32; done
for a in 1 2 4 8 16 32 12 15; do DNMF_KLUHT_VAR=410 DNMF_KLUHT_ABL=$a run 32768 16384 32; done
for v in 0 210 300; do DNMF_KLUHT_VAR=$v run 32768 16384 64; done
for a in 1 2 4 8 12 15; do DNMF_KLUHT_VAR=310 DNMF_KLUHT_ABL=$a run 32768 16384 64; done
for v in 0 100; do DNMF_KLUHT_VAR=$v run 32768 32768 128; done
cat $L
