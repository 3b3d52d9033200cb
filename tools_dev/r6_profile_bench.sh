#!/bin/bash
# Round-6 profiles of the bench itself, from the library in this tree (TAG = file prefix, default r06):
#   gpurun_out/${TAG}_stream_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --no-train-line` (inference kernels only)
#   gpurun_out/${TAG}_frame_timeline_{b1,c3}.txt  per-dispatch timelines (latency mode)
#   gpurun_out/${TAG}_train_step_kernels.txt    one steady-state training step per kernel
#   gpurun_out/${TAG}_bench.json                the default bench line, unprofiled
export TMPDIR=/tmp
TAG=${TAG:-r06}
R=$GRAFT_REPO_ROOT
D=/tmp/prof_bench; rm -rf $D; mkdir -p $D
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stream -- python3 $R/bench.py --no-train-line > $D/stream.json 2> $D/stream.err
cd $R; python tools_dev/kstats.py $(ls $D/stream/*/*kernel_stats.csv | tail -1) --top 60 --out gpurun_out/${TAG}_stream_kernel_stats.csv | head -12
B1=1 TAG=$TAG bash tools_dev/r6_timeline.sh | tail -4
D2=/tmp/prof_train; rm -rf $D2; mkdir -p $D2
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D2 -- python3 $R/bench.py --train --steps 6 --warmup 3 > $D2/log.txt 2>&1
cd $R; python tools_dev/train_step_stats.py $(ls $D2/*/*kernel_trace.csv | tail -1) 60 > gpurun_out/${TAG}_train_step_kernels.txt; head -8 gpurun_out/${TAG}_train_step_kernels.txt
timeout 1000 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; python tools_dev/bench_summary.py gpurun_out/${TAG}_bench.json
