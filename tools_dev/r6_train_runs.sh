#!/bin/bash
# three stand-alone runs of the training bench: step time, phases, and the convolution forms the per-shape autotuner settled on
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout 300 python - <<'PY'
import json, sys, io, contextlib, collections
sys.argv = ["bench.py", "--train", "--steps", "12", "--warmup", "3"]
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
from fastposecnn_amd.lib import train_conv
codes = collections.Counter(train_conv._plan_cache.values())
print(d["value"], d["ms_per_step"], d.get("phases_ms") or d["config"].get("phases_ms"), dict(codes))
PY
done
