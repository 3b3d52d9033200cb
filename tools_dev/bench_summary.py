"""Short view of a bench.py JSON line: python tools_dev/bench_summary.py <file>"""
import json, sys
l = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", l["value"], l["unit"], "ms/step", l["ms_per_step"], "repeats", l.get("repeats"), "min/max", l.get("ms_per_step_min_max"))
print("one-in-flight ms", l["config"].get("ms_per_frame_one_in_flight", l["config"].get("ms_per_step_one_in_flight")), "host-fed", l["config"].get("img_per_s_from_host_u8_frames"), "png", l["config"].get("img_per_s_from_png_files"))
for k in ("roofline", "roofline_hn128"):
    if k in l:
        r = l[k]
        print(k, "frac", r["frac"], "launch_ms", r["launch_ms"], "achieved", r["achieved"], "traffic", r.get("traffic"), "from", (r.get("from_profile") or {}).get("file"))
if "post_network" in l:
    for t in ("b1", "b32"):
        p = l["post_network"][t]
        print("post", t, "cc", p["cc"]["us"], "us", p["cc"]["frac"], "| agg", p["aggregate"]["us"], "us", p["aggregate"]["frac"])
if "backbone" in l:
    print("backbone frac", l["backbone"]["frac"], "ms", l["backbone"]["ms"], "TF", l["backbone"]["achieved"])
if "cpu_baseline" in l:
    print("cpu", l["cpu_baseline"]["value"], "cores", l["cpu_baseline"]["cores"], "|", l["cpu_baseline"]["sample"][-170:])
if "train" in l:
    print("train", {k: l["train"].get(k) for k in ("value", "ms_per_step", "error")}, l["train"].get("stages_ms"))
c2 = l.get("configs", {}).get("config2")
if c2:
    print("config2", c2["value"], "img/s; one-in-flight ms", c2["config"].get("ms_per_frame_one_in_flight"), "host-fed", c2["config"].get("img_per_s_from_host_u8_frames"),
          "png", c2["config"].get("img_per_s_from_png_files"), "vote frac", c2["roofline"]["frac"], "backbone ms", c2.get("backbone", {}).get("ms"), "frac", c2.get("backbone", {}).get("frac"))
c3 = l.get("configs", {}).get("config3")
if c3:
    print("config3", c3["value"], "img/s", "vote", c3["roofline"]["launch_ms"], "ms frac", c3["roofline"]["frac"], "backbone", c3.get("backbone", {}).get("frac"))

if l.get("plain_f32_products"): print("plain f32 products", l["plain_f32_products"]["value"], "img/s", l["plain_f32_products"].get("backbone"))
