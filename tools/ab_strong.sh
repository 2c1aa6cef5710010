#!/bin/bash
# ON THE GPU BOX: rank 0's share of the C2 frame at world sizes 1 / 8 for every mallie_amd/ab/*.so and the default library
cd "$GRAFT_REPO_ROOT" || exit 1
for lib in mallie_amd/libmallie_mgpu.so mallie_amd/ab/*.so; do
  echo "$(basename $lib .so): $(MALLIE_MGPU_LIB=$lib timeout 120 python tools/perf_strong.py 2>&1 | grep -E "world (1|8):" | tr '\n' ' ')"
done
