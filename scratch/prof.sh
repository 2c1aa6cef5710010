cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r1b
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1b/trace -o r1b -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r1b/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r1b/pmc_fetch -o r1b -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r1b/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r1b/pmc_write -o r1b -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r1b/bench_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/r1b/pmc_sq -o r1b -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r1b/bench_sq.log 2>&1
find gpurun_out/r1b -name "*.csv" | head -20
grep -h '^{' gpurun_out/r1b/bench_trace.log | cut -c1-300
