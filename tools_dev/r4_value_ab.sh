#!/bin/bash
# A/B of two builds of the library on ONE box (boxes differ by a few per cent): tools_dev/ab/libfpc_A.so and libfpc_B.so are
# copied over the in-tree library in turn; prints the streamed rate and the one-frame backbone latency of each run.
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${1:-2}); do for v in A B; do
  cp tools_dev/ab/libfpc_$v.so fastposecnn_amd/libfpc_hip.so
  timeout 600 python bench.py --no-train-line --no-config3 --no-hn128 --no-cpu-baseline --no-plain-f32 --no-batch-scan --tune-trials 2 > gpurun_out/ab_$v$i.json 2> gpurun_out/ab_$v$i.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/ab_$v$i.json").read().strip().splitlines()[-1])
print("build $v run $i value", d["value"], "ms/step", d["ms_per_step"], "backbone", d.get("backbone", {}).get("ms"))
PY
done; done
