#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash profiles/collect_r1.sh <tag>'): rocprofv3 kernel trace + separate PMC passes for
# FETCH_SIZE / WRITE_SIZE / SQ counters over the default bench command; raw CSVs land in gpurun_out/<tag>/ and are then
# summarised into profiles/ by profiles/summarize_csv.py.  PMC passes never share a run with --kernel-trace/--stats.
tag=${1:-r1}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o p -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > "$out/bench_trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$out/bench_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$out/bench_write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d "$out/pmc_sq" -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$out/bench_sq.log" 2>&1
grep -h '^{' "$out/bench_trace.log" | tail -1 > "$out/bench_line.json"
