"""HBM traffic and VALU occupancy of one fpc_ransac_voting_v3 enqueue from rocprofv3 PMC passes.

    for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
        rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/r1v/<first counter> -o p -- \
            python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline --vote-only
    done
    python tools_dev/vote_traffic.py gpurun_out/r1v profiles/r01_vote_traffic.json

Separate passes per the guide (MI355X_MICROARCH.md, HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE are in KB;
on gfx950 FETCH_SIZE reports half of the bytes of wide streaming reads -> doubled.  Median per kernel over the
dispatches of the run; the vote sequence = the ransac.hip kernels of one call."""
import csv, json, statistics, sys, collections
root, out = sys.argv[1], sys.argv[2]
VOTE = ("k_chunk_count", "k_chunk_kept", "k_compact", "k_hypothesis", "k_count_hi", "k_count_exact", "k_refine", "k_export_meta")

def load(sub):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{root}/{sub}/p_counter_collection.csv")):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fpc::", "")
        d[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return d, dur

f, _ = load("FETCH_SIZE"); w, _ = load("WRITE_SIZE"); q, dur = load("SQ_INSTS_VALU")
per = {}
fetch = write = 0.0
for k in sorted(set(f) | set(w)):
    if not k.startswith(VOTE):
        continue
    med = lambda dd, c: statistics.median(dd[k][c]) if dd.get(k) and dd[k].get(c) else 0.0
    e = {"FETCH_SIZE_KB": med(f, "FETCH_SIZE"), "WRITE_SIZE_KB": med(w, "WRITE_SIZE"),
         "calls_per_launch": round(len(f[k]["FETCH_SIZE"]) / max(1, len(f["k_refine"]["FETCH_SIZE"])), 2)}
    if q.get(k):
        insts, act, busy, gui = (med(q, c) for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"))
        e.update({"SQ_INSTS_VALU": insts, "SQ_ACTIVE_INST_VALU": act, "GRBM_GUI_ACTIVE": gui,
                  "us_under_pmc": round(statistics.median(dur[k]) / 1e3, 2),
                  # SQ_ACTIVE_INST_VALU: quad-cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE: cycles summed over the 8 XCDs
                  "valu_busy_frac_of_kernel": round(4.0 * act / 1024.0 / (gui / 8.0), 4) if gui else None})
    per[k] = e
    fetch += e["FETCH_SIZE_KB"] * e["calls_per_launch"]
    write += e["WRITE_SIZE_KB"] * e["calls_per_launch"]
res = {
    "what": "HBM-side traffic and vector-ALU occupancy of one fpc_ransac_voting_v3 enqueue (6 instances of the 640x480 vote-bench frame, hn=1000)",
    "how": __doc__.split("\n\n")[1].strip(),
    "correction": "FETCH_SIZE doubled (gfx950 reports half of the bytes of wide streaming reads, MI355X_MICROARCH.md); WRITE_SIZE as is; both in KB",
    "per_kernel": per,
    "fetch_KB_raw": round(fetch, 1), "write_KB": round(write, 1),
    "traffic_bytes_per_launch": int((2 * fetch + write) * 1024),
    "algorithmic_bytes_per_launch": 6 * 12 * 480 * 640,
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
