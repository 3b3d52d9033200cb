#!/bin/bash
# (belongs to the archived experiment in this directory: needs its kernels and fpc_vote_set_fuse_min_instances)
cd $GRAFT_REPO_ROOT
python -c "from fastposecnn_amd import build; build.build(extra=['-DFPC_STAMP_VOTE'])" > gpurun_out/tb.log 2>&1 || { tail gpurun_out/tb.log; exit 1; }
python tools_dev/vote_stamps.py --hn 1000 --frames 1 2>&1 | tail -14
python tools_dev/vote_stamps.py --hn 128 --frames 32 --fuse-min 100000 2>&1 | tail -14
python tools_dev/vote_stamps.py --hn 128 --frames 32 2>&1 | tail -14
python -c "from fastposecnn_amd import build; build.build()" > gpurun_out/tb.log 2>&1
for hn in 128 1000; do python tools_dev/vote_loop.py --hn $hn --frames 32 --iters 200 --sets 8 --bits --fuse-min 100000 | grep per-call; done
