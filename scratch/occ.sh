cd $GRAFT_REPO_ROOT
python scratch/perf.py 2>&1 | grep kernel
for s in grid32 teapot; do python scratch/perf_scenes.py $s 2>&1 | grep -E "kernel|parity"; done
SPP=16 python scratch/perf_scenes.py grid102 2>&1 | grep -E "kernel|parity"
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
