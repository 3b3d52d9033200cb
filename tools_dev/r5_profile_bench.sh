#!/bin/bash
# Round-5 profiles of the bench itself, from the library in this tree (TAG = file prefix, default r05):
#   gpurun_out/${TAG}_stream_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --no-train-line` (inference kernels only)
#   gpurun_out/${TAG}_frame_timeline_b1.txt     per-dispatch timeline of one ResNet18 batch-1 frame (latency mode: --no-pipeline)
#   gpurun_out/${TAG}_frame_timeline_c3.txt     per-dispatch timeline of one ResNet34 batch-32 step (config 3, latency mode)
#   gpurun_out/${TAG}_bench.json                the default bench line, unprofiled (skipped with NOBENCH=1)
export TMPDIR=/tmp
TAG=${TAG:-r05}
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/prof_bench; rm -rf $D; mkdir -p $D
LAT="--no-pipeline --no-batch-scan --no-train-line --no-hn128 --no-cpu-baseline --no-plain-f32 --tune-trials 1 --min-seconds 0"
if [ -z "$NOSTREAM" ]; then
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stream -- python3 $R/bench.py --no-train-line > $D/stream.json 2> $D/stream.err
cd $R; python tools_dev/kstats.py $(ls $D/stream/*/*kernel_stats.csv | tail -1) --top 60 --out gpurun_out/${TAG}_stream_kernel_stats.csv | head -24
fi
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D/frame -- python3 $R/bench.py --encoder resnet18 --batch 1 --steps 40 --warmup 10 --no-config3 $LAT > $D/frame.json 2> $D/frame.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D/c3 -- python3 $R/bench.py --encoder resnet34 --batch 32 --steps 6 --warmup 2 $LAT > $D/c3.json 2> $D/c3.err
cd $R
python tools_dev/frame_timeline.py $(ls $D/frame/*/*kernel_trace.csv | tail -1) --all > gpurun_out/${TAG}_frame_timeline_b1.txt; tail -16 gpurun_out/${TAG}_frame_timeline_b1.txt
python tools_dev/frame_timeline.py $(ls $D/c3/*/*kernel_trace.csv | tail -1) --all > gpurun_out/${TAG}_frame_timeline_c3.txt; tail -16 gpurun_out/${TAG}_frame_timeline_c3.txt
tail -n 3 $D/frame.err $D/c3.err
rm -rf $D      # raw traces stay on the box: gpurun copies back 64 MiB at most
if [ -z "$NOBENCH" ]; then
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; python tools_dev/bench_summary.py gpurun_out/${TAG}_bench.json
fi
