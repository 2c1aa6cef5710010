#!/bin/bash
# ON THE GPU BOX: everything rounds 5 and 6 built while GPU access was closed, in order of importance, each step bounded.
#   gpurun --timeout 3000 -- 'bash tools/r6_when_gpu_returns.sh r6final'        (short form: ... r6a short)
# -> gpurun_out/<tag>/{tests.txt, w5_ab.txt, multi_threads_*.txt, stream.txt, pmc + trace of the final library (profiles/collect_pmc.sh)}
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r6final}; out=gpurun_out/$tag; mkdir -p $out
short=${2:-}   # "short": tests, the W5 A/B on C4 / C3, the enqueue threads, the stream, a plain bench line (about 15 minutes)
echo "== 1. GPU suite"; timeout 900 python -m pytest tests -m gpu -q > $out/tests.txt 2>&1; echo "rc=$?"; grep -E "passed|failed|parity" $out/tests.txt | tail -3
echo "== 2. k_render_w5 beside k_render_sm"; timeout 900 bash tools/w5_ab.sh c4 c3 $([ -z "$short" ] && echo c5) > $out/w5_ab.txt 2>&1; cat $out/w5_ab.txt | cut -c1-220
MGPU_W5_BLOCK=320 timeout 600 bash tools/w5_ab.sh c4 c3 > $out/w5_ab_320.txt 2>&1; grep "MGPU_W5=1" $out/w5_ab_320.txt | cut -c1-220
for lib in mallie_amd/ab/w5_*.so; do for c in c4 c3; do echo "$(basename $lib .so) $c: $(MGPU_W5=1 MALLIE_MGPU_LIB=$lib timeout 300 python tools/perf_cfg.py $c 6 2>&1 | tail -1 | grep -o "median of the last [0-9]*: [0-9.]*")"; done; done > $out/w5_variants.txt 2>&1; cat $out/w5_variants.txt
echo "== 2b. staged primary rays for HBM-resident scenes (round 6, default on): kernel ms with and without"; for c in c4 c3; do for np in 0 1; do if [ $np = 1 ]; then export MGPU_NO_PRIM=1; else unset MGPU_NO_PRIM; fi; echo "$c MGPU_NO_PRIM=$np: $(timeout 300 python tools/perf_cfg.py $c 6 2>&1 | tail -1 | cut -c1-200)"; done; done > $out/prim_hbm_ab.txt 2>&1; unset MGPU_NO_PRIM; cat $out/prim_hbm_ab.txt
echo "== 3. eight ranks in one process: enqueue cost with and without a thread per member"
for th in 0 1; do MGPU_FRAME_ENQUEUE_THREADS=$th timeout 600 bash tools/perf_multi_one_gpu.sh $tag/multi_th$th > $out/multi_threads_$th.txt 2>&1; grep "ranks 8" $out/multi_threads_$th.txt | cut -c1-260; done
echo "== 4. the reference's stream: does it settle"; timeout 300 python tools/perf_stream.py > $out/stream.txt 2>&1; tail -6 $out/stream.txt | cut -c1-220
if [ -n "$short" ]; then echo "== 5. bench line (collects its own counter passes: profiles/pmc_current.json is of another build)"; timeout 1200 python bench.py > $out/bench_full.log 2>&1; cp -r gpurun_out/self_pmc $out/ 2>/dev/null; grep -h "^{" $out/bench_full.log | tail -1 > $out/bench_full.json
else echo "== 5. trace, counters and the bench line of the final library"; timeout 1500 bash profiles/collect_pmc.sh $tag c2 c3 c4 c5 > $out/collect.log 2>&1; tail -3 $out/collect.log; fi
python - $out/bench_full.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step")}, "roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac", "kernel_avg_ms")})
    for k, v in d.get("extra_configs", {}).items(): print(k, {x: v.get(x) for x in ("ms_per_frame", "kernel_ms_per_frame")}, (v.get("roofline") or {}).get("frac"))
    if "reference_stream_1080p" in d: print({k: v for k, v in d["reference_stream_1080p"].items() if k != "note"})
except Exception as e:
    print("no bench line:", e)
PY
