#!/bin/bash
# headline rate over frame streams x autotune objective (bench.py flags), one short run each
cd $GRAFT_REPO_ROOT
for ns in 3 4 5 6; do for tm in 0 1; do
  timeout 200 python bench.py --steps 200 --warmup 20 --no-train-line --no-config3 --no-hn128 --no-cpu-baseline --net-streams $ns --tune-mode $tm > gpurun_out/sm.json 2>/dev/null
  echo "streams=$ns tune=$tm: $(python tools_dev/bench_summary.py gpurun_out/sm.json | head -2 | tr '\n' ' ')"
done; done
