#!/bin/bash
# Level lanes (fpc_net_set_fork): parity tests, then the streamed rate / backbone latency with the lanes off and on,
# and the per-dispatch timeline of one batch-1 frame with the lanes on.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests/test_gpu_net.py -x -q 2>&1 | tail -5
for f in 0 1; do
  FPC_ENGINE_FORK=$f timeout 600 python bench.py --no-train-line --no-config3 --no-hn128 --no-cpu-baseline --no-plain-f32 --no-batch-scan > gpurun_out/fork$f.json 2> gpurun_out/fork$f.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/fork$f.json").read().strip().splitlines()[-1])
print("fork=$f value", d["value"], "ms/step", d["ms_per_step"], "backbone", d.get("backbone", {}).get("ms"), "frame latency", d.get("latency_ms", d.get("frame_latency_ms")))
PY
done
D=$R/gpurun_out/prof_fork; rm -rf $D; mkdir -p $D
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D/frame -- python3 $R/bench.py --steps 40 --warmup 10 --no-pipeline --no-batch-scan --no-train-line --no-config3 --no-hn128 --no-cpu-baseline --no-plain-f32 > $D/frame.json 2> $D/frame.err
cd $R
python tools_dev/frame_timeline.py $(ls $D/frame/*/*kernel_trace.csv | tail -1) --all > gpurun_out/fork_frame_timeline_b1.txt; tail -70 gpurun_out/fork_frame_timeline_b1.txt
