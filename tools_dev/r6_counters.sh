#!/bin/bash
# Round 6: every counter file bench.py's line cites, regenerated from the library in this tree and STAMPED with the commit given as
# $1 (the box has no .git: the caller passes `git rev-parse --short HEAD`).  rocprofv3 runs the program directly after `--`;
# counters in their own passes with the kernel trace only.  Outputs (copy into profiles/):
#   gpurun_out/r06_vote_bits_traffic_{b1,b32}_hn1000.json      FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU ... of the vote sequence (bit words)
#   gpurun_out/r06_vote_bits_{b1,b32}_hn1000_kernel_stats.csv  rocprofv3 --kernel-trace --stats of a 200-call launch loop
#   gpurun_out/r06_conv_pmc_{b1,c3}.json                        MFMA busy / wave-cycle split per kernel family of one forward
#   gpurun_out/r06_backbone_traffic_c3.json                     fabric-side bytes per kernel family of one config-3 forward
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C=${1:?usage: r6_counters.sh <commit> [vote|conv|traffic ...]}
shift
WHAT=${*:-vote conv traffic}
for what in $WHAT; do
case $what in
vote)
  for cfg in "1 1000 b1_hn1000 bits" "32 1000 b32_hn1000 bits" "32 128 b32_hn128 bits" "32 128 b32_hn128 f32"; do
    set -- $cfg; B=$1; HN=$2; TAG=$3; SRC=$4
    if [ $SRC = bits ]; then TAGP=bits_; BARG=bits; LOOPF=--bits; else TAGP=""; BARG=""; LOOPF=""; fi
    D=/tmp/prof_${TAGP}$TAG; rm -rf $D; mkdir -p $D
    for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
        first=${c%% *}
        cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $D/$first -o p -- python3 $R/tools_dev/vote_time.py $B $HN 20 $BARG > $D/$first.log 2>&1
    done
    cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 $R/tools_dev/vote_loop.py --hn $HN --frames $B --iters 200 --sets 8 $LOOPF > $D/stats.log 2>&1
    cd $R
    python tools_dev/vote_traffic.py $D $R/gpurun_out/r06_vote_${TAGP}traffic_$TAG.json $B $HN $C | grep "traffic_bytes_per_launch\|traffic_over\|valu_wave"
    python tools_dev/kstats.py $(ls $D/stats/*/*kernel_stats.csv | tail -1) --top 6 --out $R/gpurun_out/r06_vote_${TAGP}${TAG}_kernel_stats.csv | grep "k_vote"
    rm -rf $D
  done ;;
conv)
  P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
  P2="SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
  for cfg in "b1 resnet18 1" "c3 resnet34 32"; do
    set -- $cfg
    D=/tmp/pmc_$1; rm -rf $D; mkdir -p $D
    cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $D/pass1 -o p -- python3 $R/tools_dev/net_loop.py $2 $3 2 > $D/pass1.log 2>&1
    cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d $D/pass2 -o p -- python3 $R/tools_dev/net_loop.py $2 $3 2 > $D/pass2.log 2>&1
    cd $R && python tools_dev/conv_pmc.py $D gpurun_out/r06_conv_pmc_$1.json "$2-FPN, batch $3, 640x480, autotuned plans" $C 2>&1 | tail -24
    tail -n 2 $D/pass1.log $D/pass2.log | cut -c1-200
    rm -rf $D
  done ;;
traffic)
  D=/tmp/pmc_traffic; rm -rf $D; mkdir -p $D
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/fetch -o p -- python3 $R/tools_dev/net_loop.py resnet34 32 2 > $D/fetch.log 2>&1
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/write -o p -- python3 $R/tools_dev/net_loop.py resnet34 32 2 > $D/write.log 2>&1
  cd $R && python tools_dev/backbone_traffic.py $D gpurun_out/r06_backbone_traffic_c3.json "resnet34-FPN, batch 32, 640x480, autotuned plans" $C 2>&1 | tail -22
  rm -rf $D ;;
esac
done
