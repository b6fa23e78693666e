#!/usr/bin/env python3
"""Developer tool (CPU only): tests/test_dependency_margins.py over a range of seeds, margins appended to a log.
    python tests/tools/margins_long.py 100 300 profiles/r05_fuzz_dependency_margins.log [workers [configuration,...]]
Every (configuration, seed) is one call of the test function with the margin assertions BINDING (FUZZ_SEED set); a shortfall or a
misclassification is reported in the log line of that run ("FAILED: ...") and counted, the sweep goes on."""
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def run(arg):
    seed, name = arg
    os.environ["FUZZ_SEED"] = str(seed)
    import importlib
    import test_dependency_margins as T
    lib = run.lib if hasattr(run, "lib") else None
    if lib is None:
        import ctypes as C
        from dlsc_gc_planner_amd import abi
        lib = C.CDLL(os.path.join(ROOT, "tests", "emul", "libdlscqp_emul_trace.so"))
        vp = C.c_void_p
        lib.emul_plan_batch.argtypes = [C.POINTER(abi.Config), C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.emul_trace.argtypes = [C.c_int, vp, C.c_int]
        run.lib = lib
    lines = []

    def rec(key, value):
        lines.append(value)
    import warnings
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            T.test_dependency_thresholds_keep_their_margins(lib, name, seed, rec)
        return seed, name, (lines[-1] if lines else ""), None
    except BaseException as e:  # noqa: BLE001
        return seed, name, (lines[-1] if lines else "seed %d  %-32s" % (seed, name)), "%s: %s" % (type(e).__name__, str(e)[:300])


if __name__ == "__main__":
    lo, hi, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    workers = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emul"), "libdlscqp_emul_trace.so"])
    import test_dependency_margins as T
    names = sys.argv[5].split(",") if len(sys.argv) > 5 else T.NAMES  # (round 6: a long sweep of the thin-corner shape alone)
    jobs = [(s, n) for s in range(lo, hi) for n in names]
    bad = 0
    with mp.Pool(workers) as pool, open(out, "w") as fh:
        fh.write("# tests/tools/margins_long.py %d %d: tests/test_dependency_margins.py with the margin assertions binding, trace build of the kernel "
                 "source (round 5: column records); one line per (seed, configuration)\n" % (lo, hi))
        for seed, name, line, err in pool.imap(run, jobs, chunksize=4):
            if err:
                bad += 1
                line += "   FAILED: " + err
            fh.write(line + "\n")
            fh.flush()
        fh.write("# %d runs, %d with a shortfall or a misclassification\n" % (len(jobs), bad))
    print(len(jobs), "runs,", bad, "failed")
