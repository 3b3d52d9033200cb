#!/bin/bash
# Round-4 profiles of the bench itself, from the library in this tree:
#   gpurun_out/r04_stream_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --no-train-line` (inference kernels only)
#   gpurun_out/r04_frame_timeline_b1.txt     per-dispatch timeline of one batch-1 frame (latency mode: --no-pipeline)
#   gpurun_out/r04_post_kernel_stats.csv     connected components + aggregation alone (tools_dev/post_loop.py, 1 and 32 frames)
#   gpurun_out/r04_bench.json                the default bench line, unprofiled
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/prof_bench; rm -rf $D; mkdir -p $D
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stream -- python3 $R/bench.py --no-train-line > $D/stream.json 2> $D/stream.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D/frame -- python3 $R/bench.py --steps 40 --warmup 10 --no-pipeline --no-batch-scan --no-train-line --no-config3 --no-hn128 --no-cpu-baseline --no-plain-f32 > $D/frame.json 2> $D/frame.err
for f in 1 32; do
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/post$f -- python3 $R/tools_dev/post_loop.py --frames $f --iters 200 > $D/post$f.log 2>&1
done
cd $R
python tools_dev/kstats.py $(ls $D/stream/*/*kernel_stats.csv | tail -1) --top 60 --out gpurun_out/r04_stream_kernel_stats.csv | head -24
python tools_dev/frame_timeline.py $(ls $D/frame/*/*kernel_trace.csv | tail -1) --all > gpurun_out/r04_frame_timeline_b1.txt; tail -16 gpurun_out/r04_frame_timeline_b1.txt
for f in 1 32; do echo "post-network kernels, $f frame(s)"; python tools_dev/kstats.py $(ls $D/post$f/*/*kernel_stats.csv | tail -1) --top 12 --out gpurun_out/r04_post_b${f}_kernel_stats.csv | head -12; done
rm -rf $D      # raw traces stay on the box: gpurun copies back 64 MiB at most
timeout 900 python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err; python tools_dev/bench_summary.py gpurun_out/r04_bench.json
