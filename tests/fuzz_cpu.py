"""Manual tool: tests/fuzz_gpu.py's randomised sweep with the kernel's source built lane-serially (tests/emul) instead of the HIP
kernel -- the same batches, the same oracle, no GPU.    FUZZ_SEED0=100 python tests/fuzz_cpu.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dlsc_gc_planner_amd import abi
import oracle_lib as O, randqp, emul_lib as E
tot = bad = 0
S0 = int(os.environ.get('FUZZ_SEED0', '0'))
for name, (ck, gk) in randqp.CONFIGS.items():
    cfg = abi.make_config(**ck)
    for seed in range(3):
        n = 600 if 'M10' not in name and 'M12' not in name else 250
        b = randqp.random_batch(cfg, n, seed=1000 + seed + S0, **gk)
        ref = O.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], True)
        got = E.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"])
        sm = (got['status'] != ref['status']) | (got['goal_status'] != ref['goal_status'])
        ok = (ref['status'] == 0) & (got['status'] == 0)
        err = np.abs(got['x'] - ref['x']).max(1) * ok
        tot += n; bad += int(sm.sum() + (err > 1e-6).sum())
        print(name, seed, "status mismatches", int(sm.sum()), np.nonzero(sm)[0][:6], got['status'][sm][:6], ref['status'][sm][:6], "max x err %.2e" % err.max(), flush=True)
print("TOTAL", tot, "problems", bad)
