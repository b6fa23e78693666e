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


def hinted_order_numpy(keys):
    """numpy restatement of hinted_agent() in csrc/dlscqp.hip: agents in blocks of 64 ids; slot s -> block s % nb, rank s // nb;
    within a block by class descending (0 | 1 | 2-3 | 4-7 | ... | 64+), ties by id; -1 beyond the block's agents."""
    n = len(keys)
    nb = (n + 63) // 64
    cls = np.where(keys <= 0, 0, np.minimum(np.floor(np.log2(np.maximum(keys, 1))).astype(np.int64) + 1, 7))
    out = np.full(64 * nb, -1, np.int64)
    for b in range(nb):
        ids = np.arange(64 * b, min(64 * b + 64, n))
        ranked = ids[np.argsort(-cls[ids], kind="stable")]
        out[np.arange(len(ranked)) * nb + b] = ranked
    return out


@pytest.mark.parametrize("n_agents", [1, 63, 64, 65, 640, 8192 + 17])
def test_order_found_inside_the_kernel_is_the_block_wise_permutation(gpu_required, n_agents):
    """dlscqp_device_batch.hint_iters: the launch order without a sort.  The device function's answer for every slot
    (dlscqp_hint_order_preview_device) equals the numpy restatement: a permutation, most expensive class first in every block."""
    import ctypes as C
    opt = BatchTrajOptimizer(workloads.s64_config())
    try:
        rng = np.random.default_rng(n_agents)
        its = [np.zeros((n_agents, 4), np.int32) for _ in range(3)]
        for o in its:
            o[:, 0] = np.where(rng.random(n_agents) < 0.6, 0, rng.integers(0, 200, n_agents))
        td = [torch.from_numpy(o).cuda() for o in its]
        for steps in (1, 2, 3):
            nb = (n_agents + 63) // 64
            order = torch.full((64 * nb,), -7, dtype=torch.int32, device="cuda")
            ptrs = (C.c_void_p * steps)(*[t.data_ptr() for t in td[:steps]])
            rc = opt.lib.dlscqp_hint_order_preview_device(opt.handle, ptrs, steps, n_agents, C.c_void_p(order.data_ptr()), None)
            assert rc == 0, opt.lib.dlscqp_last_error()
            torch.cuda.synchronize()
            k = [its[0][:, 0].astype(np.int64), (3 * its[1][:, 0].astype(np.int64)) // 4, (9 * its[2][:, 0].astype(np.int64)) // 16]
            want = hinted_order_numpy(np.maximum.reduce(k[:steps]))
            got = order.cpu().numpy()
            assert np.array_equal(got, want)
            assert np.array_equal(np.sort(got[got >= 0]), np.arange(n_agents))
    finally:
        opt.close()


@pytest.mark.parametrize("n_swarms,swarm", [(1, 64), (3, 50), (40, 64)])
def test_a_launch_in_the_kernels_own_order_computes_the_same(gpu_required, n_swarms, swarm):
    cfg = workloads.s64_config()
    opt = BatchTrajOptimizer(cfg)
    try:
        wl = workloads.generate(cfg, lambda a, ro, r, s: opt.plan(a, ro, r, s, 0, True, True), n_swarms=n_swarms, swarm_size=swarm, seed=79)
        db = opt.make_device_batch(wl["agents"], wl["row_offsets"], wl["rows"], None, 0, want_iters=True)
        db.launch()
        first = db.results()
        prev = torch.from_numpy(first["iters"]).cuda()
        for pitch in (False, True):
            if pitch:
                db.use_row_pitch()  # every agent of the workload has K rows
            db.set_hint([prev])
            db.out_status.fill_(-1)
            db.out_ctrl.zero_()
            db.launch()
            second = db.results()
            assert (second["status"] >= 0).all()  # every agent was planned: the slots' answers are a permutation
            for k in ("status", "goal", "goal_status", "ctrl", "cost", "iters"):
                assert np.array_equal(first[k], second[k]), k
        # not together with a chunked launch
        with pytest.raises(Exception):
            db.launch(offset=0, count=len(wl["agents"]) // 2)
        db.set_hint(None)
        db.launch(offset=0, count=len(wl["agents"]) // 2)
    finally:
        opt.close()
