"""MFMA utilisation of the backbone's kernels from rocprofv3 PMC passes over tools_dev/net_loop.py (tools_dev/r5_conv_pmc.sh).

    python tools_dev/conv_pmc.py <dir with pass1/ pass2/> <out.json> <label> [commit]

One forward (the dispatches between the last two complete k_nchw3_to_nhwc4 markers) of each pass, summed per kernel family.
Units (MI355X_MICROARCH.md, rocprofv3 PMC): SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the chip's 1024 SIMDs
(32 per v_mfma_f32_32x32x16_bf16, 64 per v_mfma_f32_32x32x2_f32); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)
is the share of the kernel's cycles in which a SIMD's matrix pipe was busy, averaged over the SIMDs; SQ_WAVE_CYCLES / SQ_WAIT_* /
SQ_ACTIVE_INST_* count quad-cycles.  Counter passes serialise the kernels and run at a lower clock than the un-profiled
forward (the guide, DVFS): the busy SHARE is what carries over, not the microseconds."""
import collections, csv, glob, json, sys
root, out, label = sys.argv[1], sys.argv[2], sys.argv[3]
commit = sys.argv[4] if len(sys.argv) > 4 else None


def one_forward(sub):
    path = glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(path)))
    disp = collections.OrderedDict()
    for r in rows:
        d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"]), "c": {},
                                                    "t0": int(r.get("Start_Timestamp", 0) or 0), "t1": int(r.get("End_Timestamp", 0) or 0)})
        d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(disp)
    marks = [i for i in ids if "nchw3_to_nhwc4" in disp[i]["name"]]
    lo, hi = marks[-2], marks[-1]
    return [disp[i] for i in ids if lo <= i < hi]


def family(name):
    n = name.split("(")[0].replace("void ", "").replace("fpc::", "")
    return n


agg = collections.OrderedDict()
for sub in ("pass1", "pass2"):
    for d in one_forward(sub):
        f = agg.setdefault(family(d["name"]), {"dispatches": {}, "us_under_pmc": {}})
        f["dispatches"][sub] = f["dispatches"].get(sub, 0) + 1
        f["us_under_pmc"][sub] = f["us_under_pmc"].get(sub, 0.0) + (d["t1"] - d["t0"]) / 1e3
        for k, v in d["c"].items():
            key = k if sub == "pass1" or k != "GRBM_GUI_ACTIVE" else "GRBM_GUI_ACTIVE_pass2"
            f[key] = f.get(key, 0.0) + v
res = {"what": f"per-kernel-family SQ counters of ONE engine forward ({label}), rocprofv3 --pmc in two passes", "how": __doc__.split("\n\n")[2].strip(),
       "commit": commit, "families": {}}
tot_busy = tot_gui = 0.0
for name, f in agg.items():
    gui = f.get("GRBM_GUI_ACTIVE", 0.0)
    e = {"dispatches": f["dispatches"].get("pass1", 0), "us_under_pmc": round(f["us_under_pmc"].get("pass1", 0.0), 1)}
    for k in ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
              "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT"):
        if k in f:
            e[k] = f[k]
    if gui > 0:
        e["mfma_busy"] = round(f.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 1024), 4)
        e["clock_GHz_under_pmc"] = round(gui / 8 / max(f["us_under_pmc"].get("pass1", 1e-9), 1e-9) / 1e3, 3)
        if f.get("SQ_WAVE_CYCLES"):
            w = f["SQ_WAVE_CYCLES"]
            e["wave_cycle_shares"] = {"active_inst": round(f.get("SQ_ACTIVE_INST_ANY", 0) / w, 3), "wait_inst (issue stall)": round(f.get("SQ_WAIT_INST_ANY", 0) / w, 3),
                                      "wait_any (waitcnt / barrier)": round(f.get("SQ_WAIT_ANY", 0) / w, 3)}
        tot_busy += f.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        tot_gui += gui
    res["families"][name] = e
res["forward"] = {"mfma_busy": round(tot_busy / (tot_gui / 8 * 1024), 4) if tot_gui else None, "GRBM_GUI_ACTIVE": tot_gui,
                  "SQ_VALU_MFMA_BUSY_CYCLES": tot_busy,
                  "note": "all kernels of the forward, streaming kernels included: matrix-pipe busy cycles / (forward cycles x 1024 SIMDs)"}
json.dump(res, open(out, "w"), indent=1)
for name, e in sorted(res["families"].items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    print(f"{e.get('mfma_busy', 0):7.3f} busy  {e['us_under_pmc']:9.1f} us  x{e['dispatches']:<3d} {name[:70]}")
print("forward mfma_busy", res["forward"]["mfma_busy"])
