#!/bin/bash
# Round 6: per-dispatch timelines of one config-3 step (and, with B1=1, one batch-1 frame) from the library in this tree
export TMPDIR=/tmp
TAG=${TAG:-r06}
R=$GRAFT_REPO_ROOT
D=/tmp/prof_bench; rm -rf $D; mkdir -p $D
LAT="--no-pipeline --no-batch-scan --no-train-line --no-hn128 --no-cpu-baseline --no-plain-f32 --tune-trials 1 --min-seconds 0"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D/c3 -- python3 $R/bench.py --encoder resnet34 --batch 32 --steps 6 --warmup 2 $LAT > $D/c3.json 2> $D/c3.err
cd $R
python tools_dev/frame_timeline.py $(ls $D/c3/*/*kernel_trace.csv | tail -1) --all > gpurun_out/${TAG}_frame_timeline_c3.txt; tail -22 gpurun_out/${TAG}_frame_timeline_c3.txt
if [ -n "$B1" ]; then
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D/frame -- python3 $R/bench.py --encoder resnet18 --batch 1 --steps 40 --warmup 10 --no-config3 $LAT > $D/frame.json 2> $D/frame.err
cd $R
python tools_dev/frame_timeline.py $(ls $D/frame/*/*kernel_trace.csv | tail -1) --all > gpurun_out/${TAG}_frame_timeline_b1.txt; tail -22 gpurun_out/${TAG}_frame_timeline_b1.txt
fi
