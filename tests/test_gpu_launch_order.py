"""-m gpu: dlscqp_launch_order_device -- the next step's launch order from this step's iteration counts."""
import numpy as np
import pytest
import torch

from dlsc_gc_planner_amd import workloads
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_swarms", [1, 5, 40, 300])  # 300 swarms: beyond the single-workgroup kernel (two-pass sort)
def test_order_is_a_permutation_sorted_by_iterations(gpu_required, n_swarms):
    cfg = workloads.s64_config()
    opt = BatchTrajOptimizer(cfg)
    try:
        wl = workloads.generate(cfg, lambda a, ro, r, s: opt.plan(a, ro, r, s, 0, True, True), n_swarms=n_swarms, seed=77)
        db = opt.make_device_batch(wl["agents"], wl["row_offsets"], wl["rows"], None, 0, want_iters=True)
        db.launch()
        first = db.results()
        db.update_order()
        torch.cuda.synchronize()
        order = db.order.cpu().numpy()
        it = first["iters"][:, 0]
        assert np.array_equal(np.sort(order), np.arange(len(it)))
        key = np.minimum(it[order], 63)
        assert (np.diff(key) <= 0).all() and key[0] == min(it.max(), 63)
        # a launch in that order computes what the launch in id order computed
        db.launch()
        second = db.results()
        for k in ("status", "goal", "ctrl", "cost"):
            assert np.array_equal(first[k], second[k]), k
        # and the buffers alternate: the order a launch may still be reading is not the one the next call writes
        p0 = db.order.data_ptr()
        db.update_order()
        assert db.order.data_ptr() != p0
        torch.cuda.synchronize()
        assert np.array_equal(np.sort(db.order.cpu().numpy()), np.arange(len(it)))
    finally:
        opt.close()


def test_order_from_the_counts_of_three_steps(gpu_required):
    """dlscqp_launch_order_history_device: key = max(it_0, 3/4 it_1, 9/16 it_2) per agent."""
    cfg = workloads.s64_config()
    opt = BatchTrajOptimizer(cfg)
    try:
        wl = workloads.generate(cfg, lambda a, ro, r, s: opt.plan(a, ro, r, s, 0, True, True), n_swarms=6, seed=78)
        db = opt.make_device_batch(wl["agents"], wl["row_offsets"], wl["rows"], None, 0, want_iters=True)
        db.launch()
        it0 = db.results()["iters"][:, 0].astype(np.int64)
        N = len(it0)
        rng = np.random.default_rng(0)
        older = [np.zeros((N, 4), np.int32) for _ in range(2)]
        for o in older:
            o[:, 0] = rng.integers(0, 80, N)
        td = [torch.from_numpy(o).cuda() for o in older]
        db.update_order(history=td)
        torch.cuda.synchronize()
        want = np.maximum.reduce([it0, (3 * older[0][:, 0].astype(np.int64)) // 4, (9 * older[1][:, 0].astype(np.int64)) // 16])
        order = db.order.cpu().numpy()
        assert np.array_equal(np.sort(order), np.arange(N))
        key = np.minimum(want[order], 63)
        assert (np.diff(key) <= 0).all() and key[0] == min(want.max(), 63)
    finally:
        opt.close()


def test_order_rejects_null_buffers(gpu_required):
    import ctypes as C
    from dlsc_gc_planner_amd import abi
    opt = BatchTrajOptimizer(workloads.s64_config())
    try:
        assert opt.lib.dlscqp_launch_order_device(opt.handle, None, 5, None, None) == abi.ERR_INVALID_ARGUMENT
        assert opt.lib.dlscqp_launch_order_device(opt.handle, None, 0, None, None) == abi.SUCCESS
    finally:
        opt.close()
