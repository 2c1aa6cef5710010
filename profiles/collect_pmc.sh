#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash profiles/collect_pmc.sh <tag> [workloads...]'): rocprofv3 evidence for bench.py.
#   1. kernel trace + stats of the default bench command              -> gpurun_out/<tag>/trace
#   2. per workload (c2 c3 c4): separate PMC passes (never together with tracing) over tools/pmc_workload.py, restricted
#      to the render kernel: FETCH_SIZE | WRITE_SIZE | SQ counters     -> gpurun_out/<tag>/<workload>_{fetch,write,sq}
# profiles/summarize_pmc.py <tag> then writes profiles/<tag>_summary.txt and profiles/pmc_current.json (stamped with the
# sha256 of the library the passes ran on; bench.py uses the numbers only while that stamp matches).
tag=${1:-r2}; shift
wl=${@:-c2 c3 c4}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p "$out"
sha256sum mallie_amd/libmallie_mgpu.so | cut -d' ' -f1 > "$out/so_sha256.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o p -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$out/bench_trace.log" 2>&1
grep -h '^{' "$out/bench_trace.log" | tail -1 > "$out/bench_line.json"
for w in $wl; do
  for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" "sq2:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE"; do
    name=${pass%%:*}; ctr=${pass#*:}
    timeout 600 rocprofv3 --pmc $ctr --kernel-include-regex "k_render_sm" --output-format csv -d "$out/${w}_$name" -o p -- python tools/pmc_workload.py $w 3 > "$out/${w}_$name.log" 2>&1
  done
done
ls "$out"
