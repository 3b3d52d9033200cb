#!/bin/bash
# streamed rate of the default plan under bench.py's runtime knobs, one box
cd $GRAFT_REPO_ROOT
run() {
  timeout 600 python bench.py --no-train-line --no-config3 --no-hn128 --no-cpu-baseline --no-plain-f32 --no-batch-scan "$@" > gpurun_out/knob.json 2> gpurun_out/knob.err
  python - "$@" <<PY
import json, sys
d = json.loads(open("gpurun_out/knob.json").read().strip().splitlines()[-1])
print(" ".join(sys.argv[1:]) or "(default)", "-> value", d["value"], "backbone", d.get("backbone", {}).get("ms"))
PY
}
run
run --tune-mode 1
run --net-streams 5
run --net-streams 3
run --frames-in-flight 6
run
