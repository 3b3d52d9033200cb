"""ISA lint of libfpc_hip.so: no SALU write of vcc close before a v_div_fmas.

`python -m fastposecnn_amd.isa_lint [lib]`; fastposecnn_amd.build runs it after every link and fails the build on a
finding.

Why: hipcc expands an IEEE division into v_div_scale (writes vcc) ... v_div_fmas (reads vcc).  When it interleaves two
divisions the second flag is parked in an SGPR pair and comes back as `s_mov_b64 vcc, s[a:b]` right before the second
v_div_fmas.  On gfx950 that v_div_fmas was measured reading the stale vcc (csrc/common.hpp: div_ieee), so
the pattern is banned: divisions that the scheduler would pair go through div_ieee, and this lint proves none is left.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libfpc_hip.so")
OBJDUMP = os.environ.get("FPC_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
WINDOW = 8          # instructions looked at before each v_div_fmas

_SALU_VCC = re.compile(r"^s_\w+\s+vcc(_lo|_hi)?\b")
_VALU_VCC = re.compile(r"^v_\w+\s+(\S+,\s*)?vcc(_lo|_hi)?\b")     # v_cmp_* vcc, ... / v_div_scale vX, vcc, ... / v_add_co


def device_disassembly(lib=LIB):
    """Yield (kernel, [instruction text, ...]) for every function of every gfx950 code object bundled in `lib`."""
    tmp = tempfile.mkdtemp(prefix="fpc_isa_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for name in sorted(os.listdir(tmp)):
            if "gfx950" not in name:
                continue
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", os.path.join(tmp, name)], check=True,
                                 stdout=subprocess.PIPE, universal_newlines=True).stdout
            kern, body = None, []
            for line in txt.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    if kern is not None:
                        yield kern, body
                    kern, body = m.group(1), []
                    continue
                s = line.strip()
                if not s or kern is None:
                    continue
                s = s.split("//")[0].strip()
                if s:
                    body.append(s)
            if kern is not None:
                yield kern, body
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def scan(body, window=WINDOW):
    """[(distance, salu write, v_div_fmas)] of one function's instruction list."""
    out = []
    for i, ins in enumerate(body):
        if not ins.startswith("v_div_fmas"):
            continue
        for back in range(1, window + 1):
            if i - back < 0:
                break
            prev = body[i - back]
            if _SALU_VCC.match(prev):
                out.append((back, prev, ins))
                break
            if _VALU_VCC.match(prev):
                break                      # the nearest writer of vcc is a VALU one: the compiler's 4 wait states hold
    return out


def findings(lib=LIB, window=WINDOW):
    out = []
    n_div = 0
    for kern, body in device_disassembly(lib):
        n_div += sum(1 for ins in body if ins.startswith("v_div_fmas"))
        out += [(kern, b, p, i) for b, p, i in scan(body, window)]
    return out, n_div


def check(lib=LIB):
    bad, n_div = findings(lib)
    if bad:
        msg = "\n".join("  %s: `%s` %d instruction(s) before `%s`" % (k[:90], p, b, i) for k, b, p, i in bad[:40])
        raise RuntimeError("ISA lint: %d of %d v_div_fmas read a vcc that the SALU wrote just before them; route the "
                           "divisions through fpc::div_ieee (csrc/common.hpp):\n%s" % (len(bad), n_div, msg))
    return n_div


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else LIB
    bad, n = findings(lib)
    for k, b, p, i in bad:
        print("%s: `%s` %d before `%s`" % (k, p, b, i))
    print("%d v_div_fmas, %d with an SALU vcc write within %d instructions" % (n, len(bad), WINDOW))
    sys.exit(1 if bad else 0)
