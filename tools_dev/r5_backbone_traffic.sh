#!/bin/bash
# round 5: fabric-side bytes of the backbone's kernels at config 3 (FETCH_SIZE and WRITE_SIZE in their own passes).  TAG names the output.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C=$(cd $R && git rev-parse --short HEAD 2>/dev/null)
TAG=${TAG:-r05_backbone_traffic_c3}
D=$R/gpurun_out/pmc_traffic; rm -rf $D; mkdir -p $D
cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/fetch -o p -- python3 $R/tools_dev/net_loop.py resnet34 32 2 > $D/fetch.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/write -o p -- python3 $R/tools_dev/net_loop.py resnet34 32 2 > $D/write.log 2>&1
cd $R && python tools_dev/backbone_traffic.py $D gpurun_out/$TAG.json "resnet34-FPN, batch 32, 640x480, autotuned plans${NOTE:+; $NOTE}" $C 2>&1 | tail -22
rm -rf $D
