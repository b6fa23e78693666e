"""Time the half-space generator and the plan kernel of the x2048 stress batch separately (HIP events on torch's stream).
    DLSCQP_LIB=... python tests/tools/lsc_time.py"""
import sys
sys.argv = ["bench.py"]
sys.path.insert(0, ".")
import torch
import bench
ss = bench.StressStep(0, 0, 1, None)
ss.time(5, False)
gen, db, _, sel = ss.parts[0]


def timed(fn, n=50):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


print("generator ms %.4f  plan ms %.4f  rows kept %.0f" % (timed(lambda: gen.generate(ss.goals, plans=ss.plans)),
                                                          timed(lambda: db.launch(True, True)), ss.rows_kept_per_agent()))
ss.close()
