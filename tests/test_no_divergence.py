"""CPU tier (cross-compile only): the solver kernels contain no lane-divergent region except the guards of the
final per-lane global stores.  See "NO DIVERGENCE" in dlsc_gc_planner_amd/csrc/gi_core.hpp for why this is a
correctness rule of this code base and tests/flag_matrix.py for the GPU-side evidence."""
import shutil

import pytest

import divergence_check

ALLOWED = ("if (in) out_ctrl[j] = v;", "if (in) out_x[j] = v;", "if (in) out_slack_cost[s] =")


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_only_the_final_stores_are_lane_guarded():
    res = divergence_check.regions()
    adv = [k for k in res if "dlscqp_advance_kernel" in k]
    assert len(adv) == 1
    # the lane-strided loops of advance_agent carry compiler-made guards attributed to line 0: as many as the stand-alone
    # dlscqp_advance_kernel (the same function) shows bound what a plan kernel may show -- a guard the compiler adds anywhere else in the
    # kernel's wrapper (mask and atomic epilogue, helper loop, future code) comes on top of the epilogue's and is then seen
    adv_line0 = sum(n for (f, line), n in res[adv[0]].items() if f == "dlscqp.hip" and line == 0)
    assert 0 < adv_line0 <= 24, adv_line0
    res = {k: v for k, v in res.items() if k != adv[0]}
    assert len(res) == 23, sorted(res)  # the 8 instantiations of dlscqp_plan_kernel (DQ_INSTANCES), 3 one-wavefront-per-SIMD builds for small launches, 7 builds with helper wavefronts, the variant with its working set in HBM, four thread counts of dlscqp_sfc_kernel
    offenders = []
    # source range of the exit_stores lambda of plan_agent: its marker line up to the lambda's closing brace
    src_lines = open(divergence_check.os.path.join(divergence_check.CSRC, "gi_core.hpp")).read().split("\n")
    lo = next(i for i, l in enumerate(src_lines) if 'DQ_MARK("DQ_EXIT_STORES")' in l) + 1
    hi = next(i for i in range(lo, len(src_lines)) if src_lines[i].startswith("    };")) + 1
    assert hi - lo < 20
    # the kernel's epilogue behind plan_agent (dlscqp_device_batch.advance_after): advance_agent of dlscqp.hip, the code of
    # dlscqp_advance_kernel with its lane-guarded stores and lane-strided loops (whose compiler-made guards carry line 0)
    hip_lines = open(divergence_check.os.path.join(divergence_check.CSRC, "dlscqp.hip")).read().split("\n")
    adv_lo = next(i for i, l in enumerate(hip_lines) if l.startswith("__device__ __forceinline__ void advance_agent(")) + 1
    adv_hi = next(i for i in range(adv_lo, len(hip_lines)) if hip_lines[i] == "}") + 1
    assert 40 < adv_hi - adv_lo < 140
    for kernel, counts in res.items():
        if "dlscqp_plan_kernel" in kernel:
            line0 = sum(n for (f, line), n in counts.items() if f == "dlscqp.hip" and line == 0)
            assert line0 <= adv_line0, (kernel[-60:], line0, adv_line0)
        for (f, line), n in counts.items():
            if f == "exit_stores":  # lane-0 stores that every exit path issues as its last action
                continue
            if f == "dlscqp.hip" and "dlscqp_plan_kernel" in kernel and (line == 0 or adv_lo <= line <= adv_hi):
                continue
            src = divergence_check.source_line(f, line) if f == "gi_core.hpp" else "?"
            if f == "gi_core.hpp" and lo <= line <= hi:  # a lane-0 store inside the exit_stores lambda (every call is followed by return)
                continue
            if not src.startswith(ALLOWED):
                offenders.append((kernel[-40:], f, line, n, src[:80]))
    assert not offenders, offenders
