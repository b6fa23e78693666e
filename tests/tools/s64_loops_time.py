import sys, json
sys.path.insert(0, "/root/repo")
import bench
for G in (1, 2, 4):
    for graph in (False, True):
        r = bench.s64_closed_loop(0, groups=G, graph=graph)
        print("loops %d graph %s: %.4f ms per step, host %.4f ms" % (G, graph, r["ms_per_step"], r["host_ms_to_queue_a_step"]), flush=True)
