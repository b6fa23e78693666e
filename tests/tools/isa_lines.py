"""Static instruction count per source line of one kernel (cross-compile with line tables): where a kernel's code is.
    python tests/tools/isa_lines.py dlscqp_lsc_kernel [top]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "dlsc_gc_planner_amd", "csrc")
kern = sys.argv[1] if len(sys.argv) > 1 else "dlscqp_lsc_kernel"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "k.s")
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", 
                    "--cuda-device-only", "-gline-tables-only", "-I" + os.path.join(ROOT, "include"), "-S", "-o", out, "dlscqp.hip"], cwd=CSRC, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
files, cnt, valu, inside, cur = {}, collections.Counter(), collections.Counter(), False, None
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[m.group(1)] = os.path.basename(m.group(3) or m.group(2))
        continue
    if re.match(r"_Z\w*%s\w*:" % kern, l):
        inside = True
        continue
    if inside and l.startswith(".Lfunc_end"):
        break
    if not inside:
        continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(m.group(1), m.group(1)), int(m.group(2)))
        continue
    op = l.strip().split(" ")[0] if l.strip() else ""
    if op and not op.startswith((";", ".")) and not op.endswith(":"):
        cnt[cur] += 1
        if op.startswith("v_"):
            valu[cur] += 1
print("%s: %d instructions, %d VALU" % (kern, sum(cnt.values()), sum(valu.values())))
for (f, ln), n in valu.most_common(top):
    try:
        src = open(os.path.join(CSRC, f)).read().split("\n")[ln - 1].strip()
    except Exception:
        src = "?"
    print("%5d VALU %5d all  %s:%d  %s" % (n, cnt[(f, ln)], f, ln, src[:100]))
