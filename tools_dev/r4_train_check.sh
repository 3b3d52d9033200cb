#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q -m gpu > gpurun_out/tr_t.log 2>&1; rc=$?
echo "pytest rc=$rc $(tail -1 gpurun_out/tr_t.log | cut -c1-100)"
[ $rc = 0 ] || { tail -40 gpurun_out/tr_t.log | cut -c1-250; exit 1; }
python bench.py --train --steps 20 --warmup 4 > gpurun_out/r04_train.json 2> gpurun_out/r04_train.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_train.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); print(d["config"].get("convolutions", "")[-200:])
PY
