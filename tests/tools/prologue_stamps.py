#!/usr/bin/env python3
"""Diagnostic: where an agent that never iterates spends its cycles BEFORE the first pricing pass.  A -DDLSCQP_STAMPS -DDQ_PROLOGUE_STAMPS
build puts extra stamps into plan_agent's prologue, in the slots of the iteration's phases (which such an agent never reaches):
    cd dlsc_gc_planner_amd/csrc && hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off \
        -DDLSCQP_STAMPS -DDQ_PROLOGUE_STAMPS -I../../include -o libdlscqp_stamps.so dlscqp.hip qp_tables.cpp sfc_world.cpp grid_mapf.cpp
    python tests/tools/prologue_stamps.py        (bench.py's S64 batch, 32 768 agents)"""
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/profiles")
import stamps as S
os.environ["DLSCQP_LIB"] = S.LIB
import numpy as np, torch
from dlsc_gc_planner_amd import workloads
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
cfg = workloads.s64_config()
opt = BatchTrajOptimizer(cfg)
wl = workloads.generate(cfg, lambda a, ro, r, s: opt.plan(a, ro, r, s, 0, True, True), n_swarms=512, seed=2000)
db = opt.make_device_batch(wl["agents"], wl["row_offsets"], wl["rows"], None, 0, want_iters=True)
db.use_row_pitch()
N = len(wl["agents"])
st_t = torch.zeros((N, 12), dtype=torch.int64, device="cuda")
db.c.debug_stamps = st_t.data_ptr()
for _ in range(3): db.launch()
it = db.results()["iters"]
st = st_t.cpu().numpy().astype(np.float64)
zero = it[:, 0] == 0
names = {3: "entry", 4: "agent record, shape checks", 5: "row + slot loads issued", 6: "table loads issued, row scalings (waits for rows)", 7: "goal LP", 0: "exit stores of LP, rest",
         8: "decide T, tables, (x0 inputs)", 9: "x0, bounds table, slot scalings", 10: "finish LSC cache", 1: "screen", 2: "first pass", 11: "readout"}
print("zero-iteration agents: %d, mean total %.0f" % (zero.sum(), st[zero].sum(1).mean()))
for p in (3, 4, 5, 6, 7, 0, 8, 9, 10, 1, 2, 11):
    print("  %-52s %7.0f" % (names[p], st[zero, p].mean()))
opt.close()
