#!/bin/bash
# round 5: MFMA utilisation of today's backbone kernels from counters (Next 3).  Two --pmc passes per configuration,
# kernel trace only, the program directly after `--`.  Output: gpurun_out/r05_conv_pmc_{b1,c3}.json
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C=$(cd $R && git rev-parse --short HEAD 2>/dev/null)
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
for cfg in "b1 resnet18 1" "c3 resnet34 32"; do
  set -- $cfg
  D=$R/gpurun_out/pmc_$1; rm -rf $D; mkdir -p $D
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $D/pass1 -o p -- python3 $R/tools_dev/net_loop.py $2 $3 2 > $D/pass1.log 2>&1
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d $D/pass2 -o p -- python3 $R/tools_dev/net_loop.py $2 $3 2 > $D/pass2.log 2>&1
  cd $R && python tools_dev/conv_pmc.py $D gpurun_out/r05_conv_pmc_$1.json "$2-FPN, batch $3, 640x480, autotuned plans" $C 2>&1 | tail -24
  tail -n 2 $D/pass1.log $D/pass2.log | cut -c1-200
  rm -rf $D
done
