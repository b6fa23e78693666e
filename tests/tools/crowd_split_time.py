#!/usr/bin/env python3
"""Developer tool: the flown crowd's plan launch split by launch slots -- every block's R most expensive agents (helper-wavefront build)
and the others -- each part timed alone and both side by side, on the inputs of one mid-flight step.
    python tests/tools/crowd_split_time.py [R ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    import numpy as np
    import torch
    from dlsc_gc_planner_amd import workloads
    from dlsc_gc_planner_amd.harness import ClosedLoop
    cfg = workloads.x2048_config(device=0)
    N = 2048
    start, goal = workloads.swapping_pairs(cfg, N)
    sim = ClosedLoop(cfg, start, goal, radius=0.15, downwash=2.0, max_vel=1.0, max_acc=2.0, waypoint_step=0.5, device=0,
                     neighbours="device", nbr_pitch=256, corridor_margin=0.5, prune=True, order_hint=True)
    for _ in range(30):
        sim.step()
    torch.cuda.synchronize()
    b = sim.batch
    nb = (N + 63) // 64
    it = b.results()["iters"][:, 0]
    print("iterations of the step: mean %.2f max %d; agents with >= 8: %d" % (it.mean(), it.max(), int((it >= 8).sum())))
    side = torch.cuda.Stream()

    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    print("whole launch: %.4f ms" % timed(lambda: b.launch(True, True)))
    for R in [int(x) for x in sys.argv[1:]] or [1, 2, 4]:
        h = R * nb
        t_h = timed(lambda: b.launch(True, True, offset=0, count=h))
        t_l = timed(lambda: b.launch(True, True, offset=h, count=64 * nb - h))

        def both():
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            b.launch(True, True, stream=side, offset=0, count=h)
            b.launch(True, True, stream=cur, offset=h, count=64 * nb - h)
            cur.wait_stream(side)
        os.environ["X"] = "1"
        print("R = %d: %d heavy slots alone %.4f ms, the other %d alone %.4f ms, side by side %.4f ms" % (R, h, t_h, 64 * nb - h, t_l, timed(both)))
        os.environ["DLSCQP_WPA"] = "1"
