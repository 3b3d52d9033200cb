cd $GRAFT_REPO_ROOT
for g in 768 1024 1536 2048; do
  python -c "from fastposecnn_amd import build; build.build(extra=['-DFPC_SCAN_WGS=$g'])" > /dev/null 2>&1
  echo "scan grid $g: $(python tools_dev/vote_loop.py --hn 128 --frames 32 --iters 300 --sets 8 2>/dev/null | grep per-call) | bits: $(python tools_dev/vote_loop.py --hn 128 --frames 32 --iters 300 --sets 8 --bits 2>/dev/null | grep per-call | cut -c1-60)"
done
