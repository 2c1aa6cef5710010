cd $GRAFT_REPO_ROOT
for k in sm sm_lds; do MGPU_RENDER_KERNEL=$k python scratch/perf.py 2>&1 | grep -v amdgpu.ids; done
for m in 24 32 40 48 56; do UTIL=1 MGPU_RENDER_KERNEL=sm_lds MALLIE_MGPU_LIB=scratch/lib_sm$m.so python scratch/perf.py 2>&1 | grep -v amdgpu.ids; done
MGPU_RENDER_KERNEL=sm_lds timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
