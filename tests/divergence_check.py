"""TEST INFRASTRUCTURE: list the lane-divergent regions (EXEC-mask manipulation) the compiler emitted for the
solver kernels, with the source line each one comes from.

gi_core.hpp is written so that every lane executes every instruction (see "NO DIVERGENCE" there): the only
EXEC regions allowed are the guards of the final global stores.  Usage: python tests/divergence_check.py [-v]"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dlsc_gc_planner_amd", "csrc")
EXEC_OPS = ("s_and_saveexec", "s_or_saveexec", "s_andn2_saveexec", "s_xor_saveexec")


def regions(extra_flags=()):
    """{kernel name: Counter{(file, line): count}} for the plan kernels."""
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "--cuda-device-only",
               "-gline-tables-only", "-S", "-o", out, "dlscqp.hip"] + list(extra_flags)
        subprocess.run(cmd, cwd=CSRC, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    files = {}
    res = {}
    kernel = None
    cur = None
    since_mark = 10 ** 9
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[m.group(1)] = os.path.basename(m.group(3) or m.group(2))
            continue
        m = re.match(r'(_Z\w*dlscqp_(?:plan|sfc|advance)_kernel\w*):', l)
        if m:
            kernel = m.group(1)
            res[kernel] = Counter()
            continue
        if l.startswith(".Lfunc_end"):
            kernel = None
        if kernel is None:
            continue
        m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
        if m:
            cur = (files.get(m.group(1), m.group(1)), int(m.group(2)))
        if "DQ_EXIT_STORES" in l:
            since_mark = 0
        op = l.strip().split(" ")[0] if l.strip() else ""
        if op and not op.startswith((";", ".")):
            since_mark += 1
        if op == "s_endpgm":
            since_mark = 10 ** 9
        if op.startswith(EXEC_OPS):
            # the lane-0 stores of exit_stores() directly follow its marker comment
            res[kernel][("exit_stores", 0) if since_mark < 300 else cur] += 1
    return res


def source_line(file, line):
    try:
        return open(os.path.join(CSRC, file)).read().split("\n")[line - 1].strip()
    except Exception:
        return "?"


if __name__ == "__main__":
    r = regions(sys.argv[2:] if len(sys.argv) > 2 else ())
    for k, c in r.items():
        t = re.search(r"ILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(n?\d+)E", k)
        print("kernel <%s>: %d EXEC regions" % (",".join(t.groups()) if t else k, sum(c.values())))
        if "-v" in sys.argv[1:2]:
            for (f, ln), n in sorted(c.items(), key=lambda kv: (str(kv[0][0]), kv[0][1])):
                print("   %s:%d x%d   %s" % (f, ln, n, source_line(f, ln)[:110]))
