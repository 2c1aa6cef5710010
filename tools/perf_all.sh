#!/bin/bash
# ON THE GPU BOX: kernel ms of the default library (or $MALLIE_MGPU_LIB) on C2 + teapot + grid32 (+ grid102 with ALL=1), 16 spp each
cd "$GRAFT_REPO_ROOT" || exit 1
python tools/perf_c2.py 2>&1 | grep -o "kernel [0-9.]* ms.*checksum.*" | sed 's/^/c2: /'
for s in teapot grid32 ${ALL:+grid102}; do SPP=16 timeout 600 python tools/perf_scenes.py $s 2>&1 | grep -E "kernel|parity" | tr '\n' ' ' | sed "s/^/$s: /"; echo; done
