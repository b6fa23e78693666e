#!/usr/bin/env python3
"""Developer tool: VGPR liveness of one kernel of an AMDGPU assembly file, with the control flow (labels, branches) taken into account --
where the register pressure peaks, and WHAT is live there (every live register with the instruction that defined it).
    python tests/tools/isa_liveness.py file.s [kernel-name-fragment] [--at LINE]   (default: the first plan kernel)
Phases are the `; DQ_STAMP k` comments of a -DDLSCQP_ISA_MARKS compile (tests/tools/isa_phases.py)."""
import collections
import re
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
path = args[0]
frag = args[1] if len(args) > 1 else "plan_kernel"
at_line = int(sys.argv[sys.argv.index("--at") + 1]) if "--at" in sys.argv else None
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and frag in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))


def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out


NODST = ("global_store", "ds_write", "scratch_store", "buffer_store", "flat_store", "v_cmp", "v_readlane", "v_readfirstlane", "s_", "global_atomic",
         "ds_add", "ds_max", "ds_min", "ds_or", "ds_wr")
ins = []       # (line, op, defs, uses, text)
label_at = {}  # label -> index of the next instruction
phase_of = []
phase = -1
for i in range(start, end):
    t = lines[i].strip()
    m = re.match(r"; DQ_STAMP (\d+)", t)
    if m:
        phase = int(m.group(1))
        continue
    tl = t.split(";")[0].strip()  # (block labels carry comments: ".LBB0_2:   ; =>This Loop Header")
    if tl.endswith(":"):
        label_at[tl[:-1]] = len(ins)
        continue
    if not t or t.startswith((";", ".")):
        continue
    t = t.split(";")[0].strip()
    parts = t.split(None, 1)
    op = parts[0]
    ops = parts[1].split(",") if len(parts) > 1 else []
    defs, uses = [], []
    nodst = op.startswith(NODST)
    for k, o in enumerate(ops):
        r = regs(o)
        if k == 0 and not nodst:
            defs += r
        else:
            uses += r
    if op.startswith(("v_writelane", "v_mac", "v_fmac", "v_pk_fmac", "v_dot")) or "dpp" in t or "sdwa" in t:
        uses += defs
    ins.append((i + 1, op, defs, uses, t))
    phase_of.append(phase)
n = len(ins)
succ = [[] for _ in range(n)]
for k, (ln, op, d, u, t) in enumerate(ins):
    if op == "s_branch":
        tgt = t.split()[1]
        if tgt in label_at and label_at[tgt] < n:
            succ[k].append(label_at[tgt])
    elif op.startswith("s_cbranch"):
        tgt = t.split()[1]
        if tgt in label_at and label_at[tgt] < n:
            succ[k].append(label_at[tgt])
        if k + 1 < n:
            succ[k].append(k + 1)
    elif op in ("s_endpgm", "s_setpc_b64"):
        pass
    elif k + 1 < n:
        succ[k].append(k + 1)
live_in = [frozenset()] * n
changed = True
while changed:
    changed = False
    for k in range(n - 1, -1, -1):
        out = set()
        for s_ in succ[k]:
            out |= live_in[s_]
        new = frozenset((out - set(ins[k][2])) | set(ins[k][3]))
        if new != live_in[k]:
            live_in[k] = new
            changed = True
cnt = [len(s_) for s_ in live_in]
# phases are numbered by the stamp that ENDS them: the phase of an instruction = the next stamp's number; relabel
nxt = []
cur = None
lab = [None] * n
pending = []
ph_prev = phase_of[0] if n else -1
for k in range(n):
    pending.append(k)
    if k + 1 == n or phase_of[k + 1] != phase_of[k]:
        for q in pending:
            lab[q] = phase_of[k + 1] if k + 1 < n else 99
        pending = []
by = collections.defaultdict(list)
for k in range(n):
    by[lab[k]].append(cnt[k])
print("%d instructions; live VGPRs by phase (the stamp that ends it): max / mean" % n)
for ph in sorted(by, key=lambda p: (p is None, p)):
    print("  phase %-3s insts %5d  max %3d  mean %6.1f" % (ph, len(by[ph]), max(by[ph]), sum(by[ph]) / len(by[ph])))
k = max(range(n), key=lambda q: cnt[q]) if at_line is None else min(range(n), key=lambda q: abs(ins[q][0] - at_line))
print("\nlive at line %d (%s): %d registers; each with its reaching definition nearest above" % (ins[k][0], ins[k][4], cnt[k]))
rows = []
for r in sorted(live_in[k]):
    d = next((q for q in range(k - 1, -1, -1) if r in ins[q][2]), None)
    rows.append((d if d is not None else -1, r))
last = None
for d, r in sorted(rows):
    if d != last:
        print("  v%-4d <- line %6d: %s" % (r, ins[d][0] if d >= 0 else 0, ins[d][4] if d >= 0 else "(kernel entry)"))
        last = d
