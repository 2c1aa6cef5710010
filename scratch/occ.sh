cd $GRAFT_REPO_ROOT
for s in grid32 teapot; do
for w in 5 6; do MGPU_RENDER_BLOCKS_PER_CU=$w MALLIE_MGPU_LIB=scratch/lib_mw$w.so python scratch/perf_scenes.py $s 2>&1 | grep -E "kernel"; done
done
MGPU_RENDER_KERNEL=sm MGPU_RENDER_BLOCKS_PER_CU=4 MALLIE_MGPU_LIB=scratch/lib_mw4.so python scratch/perf.py 2>&1 | grep kernel
