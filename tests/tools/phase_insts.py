#!/usr/bin/env python3
"""Developer tool: EXECUTED instructions of dlscqp_plan_kernel by phase and kind, from the SQ counters.

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY \\
        --output-format csv -d gpurun_out/pi -- python3 tests/tools/phase_insts.py --run
    python3 tests/tools/phase_insts.py --read gpurun_out/pi

--run: the -DDQ_PHASE_PROFILE build (csrc/libdlscqp_phase.so) ends every agent's solve after phase k (tests/tools/phase_cost.py);
for k = 0, 1, 2 and the whole solve it launches, once each, a batch of agents that never iterate and bench.py's four batches --
the LAST 20 dispatches of the plan kernel in the trace.  --read: the counters of those dispatches per agent, and their differences
= what a phase executes.  (The static picture of the same phases: tests/tools/isa_phases.py.)"""
import argparse
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "dlsc_gc_planner_amd", "csrc")
PHASES = ((0, "agent record, rows, goal LP"), (1, "+ x0, bound table, scalings"), (2, "+ first pricing pass"), (-1, "whole solve"))


def run(swarms):
    import numpy as np
    import torch
    from dlsc_gc_planner_amd import abi, workloads
    from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
    lib = os.path.join(CSRC, "libdlscqp_phase.so")
    dev = torch.device("cuda", 0)
    cfg = workloads.s64_config(device=0)
    opt0 = BatchTrajOptimizer(cfg)
    wl = workloads.generate(cfg, lambda a, ro, r, s: opt0.plan(a, ro, r, s, 0, True, True), n_swarms=swarms, swarm_size=64, seed=2000, keep=4)
    hist = wl["history"]
    N, K = len(wl["agents"]), wl["K"]
    its = [opt0.make_device_batch(h["agents"], h["row_offsets"], h["rows"], None, 0, want_iters=True) for h in hist]
    for db in its:
        db.launch(True, True)
    iters = [db.results()["iters"] for db in its]
    del its
    sel = [(j, a) for j in range(4) for a in np.nonzero(iters[j][:, 0] == 0)[0]]
    pick = [sel[i % len(sel)] for i in range(N)]
    za = np.concatenate([hist[j]["agents"][a:a + 1] for j, a in pick])
    zr = np.concatenate([hist[j]["rows"][a * K:(a + 1) * K] for j, a in pick])
    opt0.close()
    abi._LIB = None
    os.environ["DLSCQP_LIB"] = lib
    os.environ["DLSCQP_NO_SELFTEST"] = "1"
    for ph, _ in PHASES:
        os.environ["DLSCQP_DEBUG_EXIT_PHASE"] = str(ph)
        opt = BatchTrajOptimizer(cfg)
        dbs = [opt.make_device_batch(h["agents"], h["row_offsets"], h["rows"], None, 0, want_iters=True) for h in hist]
        zdb = opt.make_device_batch(za, np.arange(N + 1) * K, zr, None, 0, want_iters=True)
        for db in dbs + [zdb]:
            db.use_row_pitch()
        torch.cuda.synchronize(dev)
        zdb.launch(True, True)
        for db in dbs:
            db.launch(True, True)
        torch.cuda.synchronize(dev)
        del dbs, zdb
        opt.close()
    print("agents per launch: %d; iterations per agent of the four batches: %.3f" % (N, float(np.mean([i[:, 0].mean() for i in iters]))))


def read(d, n_agents):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            rows += [r for r in csv.DictReader(fh) if "dlscqp_plan_kernel" in r.get("Kernel_Name", "")]
    by = {}
    for r in rows:
        by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(by)[-20:]
    assert len(ids) == 20, len(ids)
    names = sorted({k for i in ids for k in by[i]})
    print("per agent (counter of the launch / %d agents); SQ_WAVE_CYCLES and SQ_WAIT_ANY in quad-cycles x 4" % n_agents)
    print("%-44s " % "" + " ".join("%16s" % n.replace("SQ_", "") for n in names))
    prev = {"zero": None, "rot": None}
    for k, (ph, label) in enumerate(PHASES):
        grp = ids[5 * k:5 * k + 5]
        cur = {"zero": {n: by[grp[0]].get(n, 0.0) for n in names},
               "rot": {n: sum(by[i].get(n, 0.0) for i in grp[1:]) / 4.0 for n in names}}
        for tag in ("zero", "rot"):
            scale = lambda n, v: v * (4.0 if n in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY") else 1.0) / n_agents
            tot = " ".join("%16.1f" % scale(n, cur[tag][n]) for n in names)
            dif = " ".join("%16.1f" % scale(n, cur[tag][n] - (prev[tag][n] if prev[tag] else 0.0)) for n in names)
            print("%-44s %s" % ("%s, exit after %s" % ("never-iterating" if tag == "zero" else "four batches   ", label[:22]), tot))
            print("%-44s %s" % ("    this phase alone", dif))
        prev = cur


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--read", default=None)
    ap.add_argument("--swarms", type=int, default=512)
    a = ap.parse_args()
    if a.run:
        run(a.swarms)
    if a.read:
        read(a.read, 64 * a.swarms)
