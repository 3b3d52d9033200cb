#!/bin/bash
# count-kernel sensitivity sweep on the GPU box: per variant, rocprofv3 kernel stats of 100 vote calls (B=1, hn=1000)
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/sweep
mkdir -p $OUT
run() {   # name, then exported env assignments
    name=$1; shift
    for kv in "$@"; do export "$kv"; done
    rm -rf $OUT/$name; cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 /root/repo/tools_dev/vote_time.py ${B:-1} ${HN:-1000} 100 > $OUT/$name.log 2>&1
    cd /root/repo
    f=$(ls $OUT/$name/*/*kernel_stats.csv | tail -1)
    echo "== $name"; python tools_dev/kstats.py $f --top 4 | tail -4
    for kv in "$@"; do unset "${kv%%=*}"; done
}
run base
run dbg1 FPC_COUNT_DBG=1
run s4 FPC_COUNT_SLICES=4
run s8 FPC_COUNT_SLICES=8
run s16 FPC_COUNT_SLICES=16
run r2 FPC_COUNT_ROUNDS=2
run w4 FPC_COUNT_WAVES=4
run w6 FPC_COUNT_WAVES=6
