#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cc_label or class_compress or aggregate or pipeline or fullsize or config3_post or deferred or random_scenes or mask_bits" > gpurun_out/cc_t.log 2>&1; rc=$?
echo "pytest rc=$rc $(tail -1 gpurun_out/cc_t.log | cut -c1-100)"
[ $rc = 0 ] || { tail -50 gpurun_out/cc_t.log | cut -c1-220; exit 1; }
python tools_dev/post_loop.py 2>&1 | tail -12
python tools_dev/cc_time.py 2>&1 | grep frames; python tools_dev/agg_time.py 2>&1 | grep frames
