"""Per-slot instruction census of a K loop whose slots end in `; sched_barrier` (wino_w4.hip / wino128.hip):
    python tools_dev/slot_census.py file.s first_line last_line"""
import sys
lines = open(sys.argv[1]).read().split("\n")[int(sys.argv[2]) - 1:int(sys.argv[3])]
slot, cur = [], []
for l in lines:
    l = l.strip()
    if l.startswith("; sched_barrier"):
        slot.append(cur); cur = []; continue
    if not l or l.startswith(";") or l.startswith("."):
        continue
    cur.append(l.split()[0])
slot.append(cur)
tot = 0
for i, s in enumerate(slot):
    v = sum(1 for x in s if x.startswith("v_") and "mfma" not in x)
    m = sum(1 for x in s if "mfma" in x)
    mov = sum(1 for x in s if x.startswith("v_mov"))
    o = [x for x in s if not x.startswith("v_")]
    issue = v + sum(1 for x in o if x.startswith(("ds_", "buffer_", "global_", "s_nop")))
    tot += max(32, 8 * m + 4 * issue)
    print(f"{i:3d} mfma {m} valu {v:2d} (mov {mov}) issue-cost {8 * m + 4 * issue:3d}  {' '.join(o)}")
print("sum of max(32, 8 + 4 x single-issue instructions) over the slots:", tot)
