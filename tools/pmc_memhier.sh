#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/pmc_memhier.sh <tag> [workloads...]'): memory-hierarchy counters of the render kernel
# (rocprofv3 --pmc, separate passes, never together with tracing) over tools/pmc_workload.py: vector L1 (TCP) accesses, misses
# to L2 and their summed latency; L2 (TCC) hits / misses / requests; L2's requests to the fabric, split DRAM / other.
#                                                                      -> gpurun_out/<tag>/<workload>_{tcp,tcp2,tcc,tcc2}
# python profiles/summarize_memhier.py <tag> then writes profiles/<tag>_memhier.txt / .json.
tag=${1:-r5mh}; shift
wl=${@:-c3 c4 c5}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p "$out"
sha256sum ${MALLIE_MGPU_LIB:-mallie_amd/libmallie_mgpu.so} | cut -d' ' -f1 > "$out/so_sha256.txt"
for w in $wl; do
  for pass in "tcp:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
              "tcp2:TCP_TOTAL_READ_sum TCP_GATE_EN1_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" \
              "tcc:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
              "tcc2:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum" \
              "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAIT_INST_ANY" \
              "grbm:GRBM_GUI_ACTIVE GRBM_COUNT"; do
    name=${pass%%:*}; ctr=${pass#*:}
    timeout 900 rocprofv3 --pmc $ctr --kernel-include-regex "k_render_sm" --output-format csv -d "$out/${w}_$name" -o p -- python tools/pmc_workload.py $w 3 > "$out/${w}_$name.log" 2>&1
    echo "$w $name rc=$? $(tail -1 $out/${w}_$name.log | cut -c1-120)"
  done
done
python profiles/summarize_memhier.py $tag 2>&1 | tail -40
