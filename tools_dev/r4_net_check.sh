#!/bin/bash
# backbone change check: network parity tests, then the streamed rate / backbone latency of the default plan
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_net.py -x -q > gpurun_out/net_pytest.log 2>&1; tail -4 gpurun_out/net_pytest.log | cut -c1-300
timeout 600 python bench.py --no-train-line --no-config3 --no-hn128 --no-cpu-baseline --no-plain-f32 --no-batch-scan > gpurun_out/net_bench.json 2> gpurun_out/net_bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/net_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "backbone", d.get("backbone", {}).get("ms"))
PY
D=$GRAFT_REPO_ROOT/gpurun_out/prof_net; rm -rf $D; mkdir -p $D
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D/frame -- python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-pipeline --no-batch-scan --no-train-line --no-config3 --no-hn128 --no-cpu-baseline --no-plain-f32 > $D/frame.json 2> $D/frame.err
cd $GRAFT_REPO_ROOT
python tools_dev/frame_timeline.py $(ls $D/frame/*/*kernel_trace.csv | tail -1) --all > gpurun_out/net_frame_timeline_b1.txt; grep -n "wino\|sum of" gpurun_out/net_frame_timeline_b1.txt | head -20
