#!/bin/bash
# experiment: frame streams on disjoint shares of every XCD's CUs (FPC_CU_PARTITION=1) against the shared-chip default
cd $GRAFT_REPO_ROOT
run() {
  env $1 timeout 600 python bench.py --no-train-line --no-config3 --no-hn128 --no-cpu-baseline --no-plain-f32 --no-batch-scan $2 > gpurun_out/cupart.json 2> gpurun_out/cupart.err
  python - "$1" "$2" <<PY
import json, sys
try:
    d = json.loads(open("gpurun_out/cupart.json").read().strip().splitlines()[-1])
    print(" ".join(sys.argv[1:]), "value", d["value"], "ms/step", d["ms_per_step"], "backbone", d.get("backbone", {}).get("ms"), "one-in-flight", d.get("config", {}).get("ms_per_frame_one_in_flight"))
except Exception as e:
    print(" ".join(sys.argv[1:]), "failed", e); print(open("gpurun_out/cupart.err").read()[-1500:])
PY
}
run "FPC_CU_PARTITION=1 FPC_BENCH_SIDE_STREAM=1" "--frames-in-flight 4"
run "FPC_CU_PARTITION=1 FPC_BENCH_SIDE_STREAM=1" "--frames-in-flight 8"
run "FPC_CU_PARTITION=1 FPC_BENCH_SIDE_STREAM=1" "--net-streams 8"
run "FPC_CU_PARTITION=1 FPC_BENCH_SIDE_STREAM=1 GPU_MAX_HW_QUEUES=16" "--net-streams 8"
run "FPC_CU_PARTITION=0" ""
