export DNMF_LIB_PATH=tools/_build/libdnmf_hip_tune.so
for d in 0 1 2 4 6 7; do echo -n "dbg=$d: "; DNMF_SMALL_DBG=$d python tools/dbg/fitgap.py mu kl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['rep2']['wall_ms'])"; done
