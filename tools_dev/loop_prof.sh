#!/bin/bash
# kernel durations of the vote in a tight launch loop (GPU continuously busy): rocprofv3 stats of tools_dev/vote_loop.py
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/loop
mkdir -p $OUT
run() {
    name=$1; shift
    for kv in "$@"; do export "$kv"; done
    rm -rf $OUT/$name; cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 /root/repo/tools_dev/vote_loop.py --hn ${HN:-1000} --frames ${FR:-1} --iters 300 > $OUT/$name.log 2>&1
    cd /root/repo
    f=$(ls $OUT/$name/*/*kernel_stats.csv | tail -1)
    echo "== $name: $(grep per-call $OUT/$name.log)"; python tools_dev/kstats.py $f --top 4 | tail -4
    for kv in "$@"; do unset "${kv%%=*}"; done
}
run base
run s16 FPC_COUNT_SLICES=16
run w6 FPC_COUNT_WAVES=6
run w4 FPC_COUNT_WAVES=4
