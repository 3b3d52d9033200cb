"""Per-kernel instruction statistics of a hipcc -S --cuda-device-only listing.
    python tools_dev/isa_stats.py /tmp/nk.s [name-substring]
Prints MFMA / other vector / scratch / LDS-DMA counts and the register & LDS footprint."""
import re, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm(.*?)\.end_amdhsa_kernel', s, re.S | re.M):
    name, body, meta = m.group(1), m.group(2), m.group(3)
    if flt not in name:
        continue
    lines = [l.strip() for l in body.split('\n')]
    nm = sum(l.startswith('v_mfma') for l in lines)
    nv = sum(l.startswith('v_') and not l.startswith('v_mfma') for l in lines)
    sc = sum(l.startswith('scratch_') for l in lines)
    dma = sum(l.startswith('global_load_lds') for l in lines)
    g = lambda k: (re.search(k + r'\s+(\d+)', meta) or [0, '?'])[1]
    print(f"{name}: mfma {nm} valu {nv} scratch {sc} lds-dma {dma} vgpr {g('next_free_vgpr')} agpr {g('accum_offset')} "
          f"lds {g('group_segment_fixed_size')} priv {g('private_segment_fixed_size')}")
