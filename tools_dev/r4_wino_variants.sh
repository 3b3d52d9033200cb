#!/bin/bash
# K-loop stamps of the split-precision Winograd form for several builds (tools_dev/ab/libfpc_<V>.so), one box
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  cp tools_dev/ab/libfpc_$v.so fastposecnn_amd/libfpc_hip.so
  echo "== build $v"; python tools_dev/wino_stamps.py -5 2>&1 | grep -A6 "K loop:\|per wave" | grep -v "^--" | head -20
done
