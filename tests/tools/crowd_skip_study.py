#!/usr/bin/env python3
"""Manual tool (CPU): on the step inputs of tests/tools/crowd_dump.py, how many blocks of 64 rows a pricing pass could leave out when
it keeps, per block, the smallest scaled slack it saw and the length of the path x has gone since -- counted by the scalar solver of
tests/native built with -DSKIP_STUDY.
    g++ -O2 -march=x86-64-v3 -fPIC -std=c++17 -ffp-contract=off -DSKIP_STUDY -shared -o /tmp/study/libstudy.so \
        tests/native/native_qp.cpp dlsc_gc_planner_amd/csrc/qp_tables.cpp -lpthread
    python tests/tools/crowd_skip_study.py gpurun_out/crowd_step.npz"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dlsc_gc_planner_amd import abi, workloads  # noqa: E402


def main():
    d = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "crowd_step.npz"))
    L = C.CDLL(os.environ.get("STUDY_LIB", "/tmp/study/libstudy.so"))
    vp = C.c_void_p
    L.native_plan_batch.argtypes = [C.POINTER(abi.Config), C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    L.native_study.argtypes = [C.c_int, C.c_int, vp]
    cfg = workloads.x2048_config()
    recs = np.ascontiguousarray(d["recs"], dtype=abi.AGENT_DTYPE)
    rows = np.ascontiguousarray(d["rows"], dtype=abi.ROW_DTYPE)
    off = np.ascontiguousarray(d["rows_off"], dtype=np.int64)
    boxes = np.ascontiguousarray(d["boxes"], dtype=np.float32)
    n_top = len(recs) // 2
    for name, sel in (("most iterations", slice(0, n_top)), ("random", slice(n_top, len(recs)))):
        idx = np.arange(len(recs))[sel]
        for first, gran in ((3, 1), (2, 2), (4, 4), (8, 8)):
            L.native_study(first, gran, None)
            st = np.zeros(len(idx), np.int32)
            it = np.zeros((len(idx), 4), np.int32)
            gs = np.zeros(len(idx), np.int32)
            goal = np.zeros((len(idx), 3), np.float32)
            o2 = np.concatenate([[0], np.cumsum(np.diff(off)[idx])]).astype(np.int64)
            r2 = np.concatenate([rows[off[i]:off[i + 1]] for i in idx])
            a2 = np.ascontiguousarray(recs[idx])
            b2 = np.ascontiguousarray(boxes[idx])
            rc = L.native_plan_batch(C.byref(cfg), len(idx), a2.ctypes.data, o2.ctypes.data, r2.ctypes.data, b2.ctypes.data,
                                     0, 1, 1, goal.ctypes.data, gs.ctypes.data, None, None, None, None, st.ctypes.data, it.ctypes.data)
            assert rc == 0
            out = np.zeros(6)
            L.native_study(3, 1, out.ctypes.data)
            passes, blocks, priced, bl, pl, ag = out
            print("%-16s first block %d, %3d rows a block: iterations mean %.2f (GPU %.2f) max %d; passes %.0f, blocks priced %.0f of %.0f (%.1f %%); from the "
                  "third pass on %.0f of %.0f (%.1f %%)" % (name, first, 64 * gran, it[:, 0].mean(), d["iters"][idx, 0].mean(), it[:, 0].max(), passes, priced, blocks,
                                                             100 * priced / max(blocks, 1), pl, bl, 100 * pl / max(bl, 1)))


if __name__ == "__main__":
    main()
