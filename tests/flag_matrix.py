"""TEST INFRASTRUCTURE (manual, needs a GPU and hipcc on the box): rebuild the solver library under a list of
compiler configurations that perturb scheduling / register allocation and check every build against the
golden vectors.  A kernel whose results depend on such flags has a latent codegen problem (this is how the
lane-divergence rule of gi_core.hpp was found and verified: before it about half of these builds gave wrong
verdicts for the two-register generic instantiation, after it none).

    python tests/flag_matrix.py            # only the generic two-register instantiation (4 s per build)
    python tests/flag_matrix.py --full     # all instantiations, checks every golden batch (30 s per build)
"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dlsc_gc_planner_amd", "csrc")
# Notes.  "-mllvm -disable-machine-cse" makes this compiler emit s_mov_b64 with 64-bit literals (1.5 came out as
# s_mov_b64 s[2:3], 0x3ff8000000000000, which gfx950 cannot encode: the low half, 0, is what the hardware gets), so
# the kernel avoids non-inline double constants on scalar paths (f + 0.5 f instead of 1.5 f).  "-mllvm -sgpr-regalloc=fast" gave
# wrong verdicts in the generic instantiations in round 1; on the round-2 source it is clean (all twelve golden batches, default
# and forced-generic dispatch) and is part of the list.  Under it the compiler emits ~400 whole-wave-mode save / restore
# regions around the spills of the VGPRs that carry spilled SGPRs (tests/divergence_check.py -v -mllvm -sgpr-regalloc=fast); a scan
# of the generated code found no SCC or EXEC value clobbered across them, and the round-1 failure could not be reproduced to
# bisect it.  "-mllvm -vgpr-regalloc=fast" (4 KB of scratch per lane) faulted in two batches in round 1 and is not in the list.
FLAGS = ["", "-O2", "-O1", "-fno-unroll-loops",
         "-mllvm -amdgpu-waitcnt-forcezero=1", "-mllvm -amdgpu-dpp-combine=false", "-mllvm -enable-misched=false",
         "-mllvm -enable-post-misched=false", "-mllvm -amdgpu-load-store-vectorizer=false",
         "-mllvm -amdgpu-scalarize-global-loads=false", "-mllvm -amdgpu-sdwa-peephole=false",
         "-mllvm -amdgpu-opt-exec-mask-pre-ra=false", "-mllvm -disable-machine-licm", "-mllvm -disable-machine-sink",
         "-mllvm -disable-machine-cse", "-mllvm -disable-peephole", "-mllvm -disable-branch-fold",
         "-mllvm -vgpr-regalloc=basic", "-mllvm -sgpr-regalloc=basic", "-mllvm -sgpr-regalloc=fast",
         "-mllvm -amdgpu-enable-rewrite-partial-reg-uses=false", "-mllvm -amdgpu-use-aa-in-codegen=false",
         "-mllvm -amdgpu-opt-vgpr-liverange=false", "-mllvm -split-spill-mode=size", "-mllvm -split-spill-mode=speed",
         "-mllvm -greedy-reverse-local-assignment", "-mllvm -join-liveintervals=false",
         # scheduling strategies; iterative-ilp is the one the product is built with (__graft_entry__.HIPCC_FLAGS)
         "-mllvm -amdgpu-sched-strategy=iterative-ilp", "-mllvm -amdgpu-sched-strategy=max-ilp",
         "-mllvm -amdgpu-sched-strategy=iterative-maxocc", "-mllvm -amdgpu-sched-strategy=iterative-minreg"]

CHECK = r'''
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np, fixtures
from dlsc_gc_planner_amd import abi
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
names = %(names)r
out = []
for name in names:
    cfg, b, ans = fixtures.load(name)
    opt = BatchTrajOptimizer(cfg)
    db = opt.make_device_batch(b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], want_f64=True, want_iters=True)
    db.launch()
    got = db.results()
    reached = ans["qp_status"] >= 0
    bad = int((reached & ((got["status"] != abi.OK) != (ans["qp_status"] != abi.OK))).sum())
    ok = reached & (ans["qp_status"] == abi.OK) & (got["status"] == abi.OK)
    err = float(np.abs(got["x"][ok] - ans["x"][ok]).max()) if ok.any() else 0.0
    out.append("%%s:%%d:%%.3e" %% (name.split("_")[0], bad, err))
    opt.close()
print("RESULT", " ".join(out))
'''


def main():
    full = "--full" in sys.argv
    flags_list = FLAGS
    if "--flags" in sys.argv:  # one explicit configuration
        flags_list = [sys.argv[sys.argv.index("--flags") + 1]]
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fixtures
    names = list(fixtures.NAMES) if full else ["forest3d_M10_dyn", "long3d_M12_comm"]
    n_bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for i, flags in enumerate(flags_list):
            so = os.path.join(tmp, "v%d.so" % i)
            cmd = "hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off %s %s -o %s dlscqp.hip qp_tables.cpp sfc_world.cpp grid_mapf.cpp" % (
                "" if full else "-DDLSCQP_ONLY_G2", flags, so)
            t0 = time.time()
            p = subprocess.run(cmd, shell=True, cwd=CSRC, capture_output=True, text=True)
            if p.returncode != 0:
                print("[%2d] %-60s compile failed" % (i, flags))
                continue
            q = subprocess.run([sys.executable, "-c", CHECK % dict(root=ROOT, names=names)],
                               env=dict(os.environ, DLSCQP_LIB=so), capture_output=True, text=True)
            r = [l for l in q.stdout.splitlines() if l.startswith("RESULT")]
            line = r[0] if r else "RUN FAILED " + (q.stderr or "")[-300:]
            # tokens are name:wrong-verdicts:max|x - golden|
            wrong = (not r) or any(tok.split(":")[1] != "0" or float(tok.split(":")[2]) > 5e-6 for tok in line.split()[1:])
            n_bad += wrong
            print("[%2d] %-60s %s %s (%.0f s)" % (i, flags or "(default)", "WRONG" if wrong else "ok   ", line, time.time() - t0), flush=True)
    print("%d of %d configurations wrong" % (n_bad, len(flags_list)))
    return 1 if n_bad else 0


if __name__ == "__main__":
    sys.exit(main())
