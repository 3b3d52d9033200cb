#!/bin/bash
# streamed rate of the default plan under runtime knobs, one box
cd $GRAFT_REPO_ROOT
run() {
  env $1 timeout 600 python bench.py --no-train-line --no-config3 --no-hn128 --no-cpu-baseline --no-plain-f32 --no-batch-scan $2 > gpurun_out/knob.json 2> gpurun_out/knob.err
  python - "$1" "$2" <<PY
import json, sys
try:
    d = json.loads(open("gpurun_out/knob.json").read().strip().splitlines()[-1])
    print(" ".join(sys.argv[1:]) or "(default)", "-> value", d["value"], "backbone", d.get("backbone", {}).get("ms"), "trials", d["config"].get("trial_rates_img_per_s"))
except Exception as e:
    print("failed", e); print(open("gpurun_out/knob.err").read()[-2000:])
PY
}
run "FPC_X=1" ""
run "FPC_X=1" "--frames-in-flight 6"
run "FPC_X=1" "--frames-in-flight 8"
run "FPC_X=1" "--frames-in-flight 4"
run "FPC_X=1" ""
