"""Static instruction mix of one plan-kernel instantiation by phase of plan_agent.  A diagnostic compile (-DDLSCQP_ISA_MARKS) turns
every w.stamp(k) of gi_core.hpp -- the phase boundaries profiles/stamps.py times on the device -- into a comment in the assembly;
the instructions in front of a marker belong to the phase it ends.  By kind: FP64 VALU (4 cycles per wave64 instruction), other
VALU (2), SALU, LDS, VMEM, SMEM, waits.  STATIC counts: a loop body counts once however often it runs (the loops over the
active rows' records run ceil(q / 8) and ceil(q / 4) times), so this is the code's shape, not the executed stream; the executed
totals come from the SQ_INSTS_* counters (profiles/collect.sh).
    python tests/tools/isa_phases.py [mangled-name-fragment]   (default: the S64 instantiation)"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "dlsc_gc_planner_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "profiles"))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
frag = args[0] if args else "plan_kernelILi1ELi3ELi13ELi5ELi2ELi0ELi0ELi0ELb0ELi1"
asm = "/tmp/isa_phases.s"
if "--reuse" not in sys.argv or not os.path.exists(asm):
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", 
                    "--cuda-device-only", "-DDLSCQP_ISA_MARKS", "-I" + os.path.join(ROOT, "include"), "-S", "-o", asm, "dlscqp.hip"], cwd=CSRC, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and frag in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))


def kind(op):
    if op.startswith("v_"):
        return "valu64" if "f64" in op else "valu32"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_load", "s_buffer")):
        return "smem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


from stamps import PHASES  # noqa: E402  (the same names the device stamps are printed under)
names = list(PHASES) + ["(behind the last marker: exits)"]
tot = [collections.Counter() for _ in names]
pend = collections.Counter()
for l in lines[start:end]:
    t = l.strip()
    m = re.match(r"; DQ_STAMP (\d+)", t)
    if m:
        tot[int(m.group(1))].update(pend)
        pend = collections.Counter()
        continue
    if not t or t.startswith((";", ".")) or t.endswith(":"):
        continue
    pend[kind(t.split()[0])] += 1
tot[-1].update(pend)
KINDS = ["valu64", "valu32", "salu", "lds", "vmem", "smem", "wait", "other"]
print("%-34s %6s  " % (frag[-34:], "all") + " ".join("%7s" % k for k in KINDS) + "   VALU issue cycles (4 x FP64 + 2 x other)")
allc = collections.Counter()
for name, c in zip(names, tot):
    allc.update(c)
    print("%-34s %6d  " % (name, sum(c.values())) + " ".join("%7d" % c[k] for k in KINDS) + "   %6d" % (4 * c["valu64"] + 2 * c["valu32"]))
print("%-34s %6d  " % ("total (static)", sum(allc.values())) + " ".join("%7d" % allc[k] for k in KINDS) + "   %6d" % (4 * allc["valu64"] + 2 * allc["valu32"]))
