"""TEST INFRASTRUCTURE (manual, GPU box): bench.py under a list of compiler flag sets (one per line of the file\ngiven as argv[1]); prints ms/step, QP/s and kernel time per build."""
import subprocess, sys, os, time, json
src = "/root/repo/dlsc_gc_planner_amd/csrc"
V = [l.rstrip("\n") for l in open(sys.argv[1]) if l.strip() and not l.startswith("#")]
for i, flags in enumerate(V):
    so = "/tmp/p%d.so" % i
    cmd = "cd %s && hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off %s -o %s dlscqp.hip qp_tables.cpp sfc_world.cpp" % (src, flags, so)
    p = subprocess.run(cmd, shell=True, capture_output=True, text=True)
    if p.returncode != 0:
        print("[%2d] %-70s COMPILE-FAIL %s" % (i, flags, [l for l in p.stderr.splitlines() if "error" in l][:1])); continue
    q = subprocess.run([sys.executable, "/root/repo/bench.py", "--no-cpu-baseline", "--no-latency", "--steps", "30"], env=dict(os.environ, DLSCQP_LIB=so), capture_output=True, text=True)
    try:
        j = json.loads(q.stdout.strip().splitlines()[-1])
        print("[%2d] %-70s %.4f ms/step  %.1f M QP/s  kernel %.4f ms  solved %.4f" % (i, flags, j["ms_per_step"], j["value"] / 1e6, j["roofline"]["avg_kernel_ms"], j["config"]["solved_fraction"]), flush=True)
    except Exception as e:
        print("[%2d] %-70s RUN-FAIL %s" % (i, flags, (q.stderr or "")[-300:]))
