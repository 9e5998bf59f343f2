cd $GRAFT_REPO_ROOT
export DNMF_LIB_PATH=tools/_build/libdnmf_hip_tune.so
mkdir -p gpurun_out/r04d; L=gpurun_out/r04d/updgrid.log; : > $L
for g in 1073741824 256 512 768 1024 2048 4096; do
  for e in mu_update_h mu_update_w; do echo "grid=$g" >> $L; DNMF_UPD_GRID=$g ELT=$e timeout 100 python tools/eltbench.py 128 2>/dev/null | grep bytes >> $L; done
done
for g in 512 1024 2048; do
  for e in mu_update_h mu_update_w; do echo "k64 grid=$g" >> $L; DNMF_UPD_GRID=$g ELT=$e timeout 100 python tools/eltbench.py 64 2>/dev/null | grep bytes >> $L; done
done
cat $L
