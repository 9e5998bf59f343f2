R=${GRAFT_REPO_ROOT:-$(pwd)}
export DNMF_LIB_PATH=$R/tools/_build/libdnmf_hip_tune.so
for h in 14 16 99; do for g in 512 1024 1536 2048 4096 100000; do
 echo "H=$h grid=$g $(DNMF_UPD_H=$h DNMF_UPD_GRID=$g ELT=mu_update_h python3 $R/tools/eltbench.py 64 | tail -1)"
done; done
for w in 4 6; do for g in 1024 2048 100000; do
 echo "W=$w grid=$g $(DNMF_UPD_W=$w DNMF_UPD_GRID=$g ELT=mu_update_w python3 $R/tools/eltbench.py 64 | tail -1)"
done; done
