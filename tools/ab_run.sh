#!/bin/bash
# ON THE GPU BOX: times every mallie_amd/ab/*.so (and the default library) on C2 / teapot / grid32.
# usage: bash tools/ab_run.sh [scenes...]   (default: c2 teapot grid32)
cd "$GRAFT_REPO_ROOT" || exit 1
scenes=${@:-c2 teapot grid32}
for lib in mallie_amd/libmallie_mgpu.so mallie_amd/ab/*.so; do
  line="$(basename $lib .so):"
  for s in $scenes; do
    if [ $s = c2 ]; then r=$(MALLIE_MGPU_LIB=$lib timeout 120 python tools/perf_c2.py 2>&1 | grep -o "kernel [0-9.]* ms");
    else r=$(MALLIE_MGPU_LIB=$lib SPP=16 timeout 300 python tools/perf_scenes.py $s 2>&1 | grep -o "kernel [0-9.]* ms"); fi
    line="$line $s $r |"
  done
  echo "$line"
done
