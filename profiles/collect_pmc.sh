#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash profiles/collect_pmc.sh <tag> [workloads...]'): rocprofv3 evidence for bench.py.
#   1. kernel trace + stats of the default bench command              -> gpurun_out/<tag>/trace
#   2. per workload (c2 c3 c4 c5): separate PMC passes (never together with tracing) over tools/pmc_workload.py, restricted
#      to the render kernel: FETCH_SIZE | WRITE_SIZE | SQ counters     -> gpurun_out/<tag>/<workload>_{fetch,write,sq}
#   3. calibration of FETCH_SIZE / WRITE_SIZE on known byte counts: profiles/microbench/pmc_calib (1 GiB streamed in the
#      library's access patterns) and k_accumulate_tiled of the c2 passes (398 MB of planes read, 24.9 MB of image written)
#                                                                      -> gpurun_out/<tag>/calib_{fetch,write}, acc_{fetch,write}
# profiles/summarize_pmc.py <tag> then writes profiles/<tag>_summary.txt and profiles/pmc_current.json (stamped with the
# sha256 of the library the passes ran on; bench.py uses the numbers only while that stamp matches).
tag=${1:-r2}; shift
wl=${@:-c2 c3 c4 c5}
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
# calibration: known byte counts (MI355X_MICROARCH.md, HBM: calibrate before trusting an absolute)
if [ ! -x profiles/microbench/pmc_calib ]; then hipcc --offload-arch=gfx950 -O3 profiles/microbench/pmc_calib.hip -o profiles/microbench/pmc_calib > "$out/calib_build.log" 2>&1; fi
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${pass%%:*}; ctr=${pass#*:}
  timeout 300 rocprofv3 --pmc $ctr --kernel-include-regex "k_calib" --output-format csv -d "$out/calib_$name" -o p -- profiles/microbench/pmc_calib > "$out/calib_$name.log" 2>&1
  timeout 600 rocprofv3 --pmc $ctr --kernel-include-regex "k_accumulate_tiled" --output-format csv -d "$out/acc_$name" -o p -- python tools/pmc_workload.py c2 3 > "$out/acc_$name.log" 2>&1
done
# the numbers above, summarised on the box, and then the full default bench line of the same library WITH them (its roofline
# objects are filled from profiles/pmc_current.json only when the stamp matches the loaded library)
python profiles/summarize_pmc.py $tag > "$out/summary_on_box.log" 2>&1
cp profiles/pmc_current.json "$out/pmc_current.json"
timeout 1200 python bench.py > "$out/bench_full.log" 2>&1
grep -h '^{' "$out/bench_full.log" | tail -1 > "$out/bench_full.json"
ls "$out"
