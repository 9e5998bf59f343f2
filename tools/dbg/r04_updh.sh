cd $GRAFT_REPO_ROOT
export DNMF_LIB_PATH=tools/_build/libdnmf_hip_tune.so
mkdir -p gpurun_out/r04d; L=gpurun_out/r04d/updh.log; : > $L
for v in 14 13 15 22 23; do for g in 512 1024 2048; do
  echo "UPD_H=$v grid=$g" >> $L; DNMF_UPD_H=$v DNMF_UPD_GRID=$g ELT=mu_update_h timeout 100 python tools/eltbench.py 128 2>/dev/null | grep bytes >> $L
done; done
cat $L
