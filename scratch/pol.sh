cd $GRAFT_REPO_ROOT
python scratch/perf.py 2>&1 | grep kernel
for v in p1 p2 p3 sm16 sm20 sm24 sm28 sm40; do MALLIE_MGPU_LIB=scratch/lib_$v.so python scratch/perf.py 2>&1 | grep kernel; done
for v in p1 p3 sm20 sm24; do MALLIE_MGPU_LIB=scratch/lib_$v.so python scratch/perf_scenes.py grid32 2>&1 | grep kernel; done
python scratch/perf_scenes.py grid32 2>&1 | grep kernel
