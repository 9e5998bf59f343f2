R=${GRAFT_REPO_ROOT:-$(pwd)}
export DNMF_LIB_PATH=$R/tools/_build/libdnmf_hip_tune.so
export DNMF_UPD_GRID=100000000
for k in 64 32 128; do
for h in 14 15 16 23 24; do
 echo "k=$k H=$h $(DNMF_UPD_H=$h ELT=mu_update_h python3 $R/tools/eltbench.py $k | tail -1)"
done
for w in 3 4 5 6; do
 echo "k=$k W=$w $(DNMF_UPD_W=$w ELT=mu_update_w python3 $R/tools/eltbench.py $k | tail -1)"
done; done
