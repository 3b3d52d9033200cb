"""Fabric-side bytes of the backbone's kernels from rocprofv3 PMC passes over tools_dev/net_loop.py (tools_dev/r5_backbone_traffic.sh).

    python tools_dev/backbone_traffic.py <dir with fetch/ write/> <out.json> <label> [commit]

One forward (between the last two complete k_nchw3_to_nhwc4 markers) of each pass, summed per kernel family.  FETCH_SIZE and
WRITE_SIZE are in KB, separate passes (MI355X_MICROARCH.md: they do not fit one pass; counters never beside system traces).
On gfx950 FETCH_SIZE reports half of the bytes of wide (16 B per lane) coalesced reads: doubled here as the guide prescribes,
which OVERSTATES kernels whose reads are narrower; ratios between variants of one kernel are unaffected.  Infinity-Cache hits are counted."""
import collections, csv, glob, json, sys
root, out, label = sys.argv[1], sys.argv[2], sys.argv[3]
commit = sys.argv[4] if len(sys.argv) > 4 else None


def one_forward(sub, counter):
    path = glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True)[0]
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "v": 0.0, "t0": int(r.get("Start_Timestamp", 0) or 0), "t1": int(r.get("End_Timestamp", 0) or 0)})
        if r["Counter_Name"] == counter:
            d["v"] += float(r["Counter_Value"])
    ids = sorted(disp)
    marks = [i for i in ids if "nchw3_to_nhwc4" in disp[i]["name"]]
    lo, hi = marks[-2], marks[-1]
    return [disp[i] for i in ids if lo <= i < hi]


fam = collections.OrderedDict()
for sub, counter, key in (("fetch", "FETCH_SIZE", "fetch_KB_raw"), ("write", "WRITE_SIZE", "write_KB")):
    for d in one_forward(sub, counter):
        n = d["name"].split("(")[0].replace("void ", "").replace("fpc::", "")
        f = fam.setdefault(n, {"dispatches": 0, "fetch_KB_raw": 0.0, "write_KB": 0.0, "us_under_pmc": 0.0})
        f[key] += d["v"]
        if sub == "fetch":
            f["dispatches"] += 1
            f["us_under_pmc"] += (d["t1"] - d["t0"]) / 1e3
res = {"what": f"fabric-side read / write bytes per kernel family of ONE engine forward ({label})", "how": __doc__.split("\n\n")[2].strip(), "commit": commit, "families": {}}
tf = tw = 0.0
for n, f in sorted(fam.items(), key=lambda kv: -(2 * kv[1]["fetch_KB_raw"] + kv[1]["write_KB"])):
    rd, wr = 2 * f["fetch_KB_raw"] * 1024, f["write_KB"] * 1024
    res["families"][n] = {"dispatches": f["dispatches"], "read_MB": round(rd / 1e6, 1), "write_MB": round(wr / 1e6, 1), "us_under_pmc": round(f["us_under_pmc"], 1),
                          "TBps_under_pmc": round((rd + wr) / max(f["us_under_pmc"], 1e-9) / 1e6, 2)}
    tf += rd; tw += wr
    print(f"{rd / 1e6:9.1f} MB read {wr / 1e6:9.1f} MB written {f['us_under_pmc']:9.1f} us x{f['dispatches']:<3d} {n[:64]}")
res["forward"] = {"read_MB": round(tf / 1e6, 1), "write_MB": round(tw / 1e6, 1)}
json.dump(res, open(out, "w"), indent=1)
print("forward", res["forward"])
