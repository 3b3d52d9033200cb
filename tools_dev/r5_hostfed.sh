#!/bin/bash
# round 5: the batch-1 host-fed rate of bench.py against the same loop in tools_dev/hostfed_breakdown.py
cd $GRAFT_REPO_ROOT
Q="--encoder resnet18 --batch 1 --no-config3 --no-batch-scan --no-train-line --no-hn128 --no-cpu-baseline --no-plain-f32 --tune-trials 1"
for nh in 0 300; do
FPC_BENCH_HOST_FRAMES=$nh timeout 300 python bench.py $Q 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=l['config']
print('host frames $nh:', 'value', l['value'], 'host-fed', c['img_per_s_from_host_u8_frames'], 'png', c['img_per_s_from_png_files'])"
done
python tools_dev/hostfed_breakdown.py 2>&1 | tail -2
