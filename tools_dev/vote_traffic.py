"""HBM traffic and VALU work of one fpc_ransac_voting_v3 enqueue from rocprofv3 PMC passes.

    cd /tmp; for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
        rocprofv3 --kernel-trace --pmc $c --output-format csv -d $REPO/gpurun_out/<dir>/<first counter> -o p -- \
            python3 $REPO/tools_dev/vote_time.py <B> <hn> 20
    done
    python tools_dev/vote_traffic.py gpurun_out/<dir> profiles/r03_vote_traffic_<tag>.json <B> <hn> <commit>

Separate passes per the guide (MI355X_MICROARCH.md, HBM / rocprofv3: FETCH_SIZE and WRITE_SIZE do not fit one pass,
and counters are never combined with the system traces).  FETCH_SIZE and WRITE_SIZE are in KB; on gfx950 FETCH_SIZE
reports half of the bytes of wide (16 B per lane) streaming reads -> doubled; WRITE_SIZE as is.  Median per kernel over
the dispatches of the run; the vote sequence = the four k_vote_* kernels of one call."""
import collections
import csv
import glob
import json
import statistics
import sys

root, out = sys.argv[1], sys.argv[2]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
hn = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
commit = sys.argv[5] if len(sys.argv) > 5 else None
VOTE = ("k_vote_scan", "k_vote_plan", "k_vote_count", "k_vote_final")


def load(sub):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    path = glob.glob(f"{root}/{sub}/**/p_counter_collection.csv", recursive=True) or glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True)
    for r in csv.DictReader(open(path[0])):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fpc::", "").split("<")[0]
        d[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return d, dur


f, _ = load("FETCH_SIZE"); w, _ = load("WRITE_SIZE"); q, dur = load("SQ_INSTS_VALU")
per = {}
fetch = write = valu = 0.0
ncalls = max(1, len(f["k_vote_final"]["FETCH_SIZE"]))
for k in VOTE:
    med = lambda dd, c: statistics.median(dd[k][c]) if dd.get(k) and dd[k].get(c) else 0.0
    e = {"FETCH_SIZE_KB": med(f, "FETCH_SIZE"), "WRITE_SIZE_KB": med(w, "WRITE_SIZE"),
         "calls_per_launch": round(len(f[k]["FETCH_SIZE"]) / ncalls, 2)}
    if q.get(k):
        insts, act, busy, gui = (med(q, c) for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"))
        e.update({"SQ_INSTS_VALU": insts, "SQ_ACTIVE_INST_VALU": act, "GRBM_GUI_ACTIVE": gui,
                  "us_under_pmc": round(statistics.median(dur[k]) / 1e3, 2)})
        valu += insts
    per[k] = e
    fetch += e["FETCH_SIZE_KB"] * e["calls_per_launch"]
    write += e["WRITE_SIZE_KB"] * e["calls_per_launch"]
n_inst = 6 * B
res = {
    "what": f"HBM-side traffic and vector-ALU work of one fpc_ransac_voting_v3 enqueue ({n_inst} instances of the 640x480 vote-bench "
            f"fixture, batch {B}, hn={hn})",
    "how": __doc__.split("\n\n")[1].strip(),
    "correction": "FETCH_SIZE doubled (gfx950 reports half of the bytes of wide streaming reads, MI355X_MICROARCH.md); WRITE_SIZE as is; both in KB",
    "per_kernel": per,
    "fetch_KB_raw": round(fetch, 1), "write_KB": round(write, 1),
    "traffic_bytes_per_launch": int((2 * fetch + write) * 1024),
    "algorithmic_bytes_per_launch": n_inst * 12 * 480 * 640,
    "valu_wave_instructions_per_launch": int(valu),
    "commit": commit,
}
res["traffic_over_algorithmic"] = round(res["traffic_bytes_per_launch"] / res["algorithmic_bytes_per_launch"], 3)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
