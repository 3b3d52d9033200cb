#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_net.py -x -q -k "level_lanes or graph_replay" > gpurun_out/fork_pytest.log 2>&1; tail -30 gpurun_out/fork_pytest.log | cut -c1-300
for g in 0; do for f in 0 1; do
  FPC_ENGINE_GRAPH=$g FPC_ENGINE_FORK=$f timeout 600 python bench.py --no-train-line --no-config3 --no-hn128 --no-cpu-baseline --no-plain-f32 --no-batch-scan > gpurun_out/fork$f.g$g.json 2> gpurun_out/fork$f.g$g.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/fork$f.g$g.json").read().strip().splitlines()[-1])
print("graph=$g fork=$f value", d["value"], "ms/step", d["ms_per_step"], "backbone", d.get("backbone", {}).get("ms"))
PY
done; done
D=$R/gpurun_out/prof_fork2; rm -rf $D; mkdir -p $D
cd /tmp && FPC_ENGINE_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D/frame -- python3 $R/bench.py --steps 40 --warmup 10 --no-pipeline --no-batch-scan --no-train-line --no-config3 --no-hn128 --no-cpu-baseline --no-plain-f32 > $D/frame.json 2> $D/frame.err
cd $R
python tools_dev/frame_timeline.py $(ls $D/frame/*/*kernel_trace.csv | tail -1) --all > gpurun_out/fork_frame_timeline_b1_plain.txt; tail -45 gpurun_out/fork_frame_timeline_b1_plain.txt
