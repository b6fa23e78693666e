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


def _classes(it):
    it = np.asarray(it)
    return np.where(it <= 0, 0, np.minimum(7, np.floor(np.log2(np.maximum(it, 1))).astype(np.int64) + 1))


def _mask_order(masks, n):
    """numpy restatement of masked_agent (csrc/dlscqp.hip): slot s = rank * n_blocks + block"""
    nb = (n + 63) // 64
    out = np.full(64 * nb, -1, np.int64)
    for b in range(nb):
        nvalid = min(64, n - 64 * b)
        bits = [[(int(masks[b, c]) >> l) & 1 for l in range(64)] for c in range(8)]
        tot = np.sum(bits, axis=0)
        ok = all(tot[l] == (1 if l < nvalid else 0) for l in range(64))
        lanes = [l for c in range(7, -1, -1) for l in range(64) if bits[c][l]] if ok else list(range(nvalid))
        for r, l in enumerate(lanes):
            out[r * nb + b] = 64 * b + l
    return out


@pytest.mark.parametrize("n_swarms,swarm_size", [(1, 64), (3, 50), (40, 64)])  # 150 agents: a ragged last block
def test_class_masks_order_the_next_launch(gpu_required, n_swarms, swarm_size):
    """dlscqp_device_batch.hint_mask_*: the masks a launch leaves behind hold every agent's bit in the class of its iteration count;
    the next launch's order from them equals the numpy restatement, plans every agent exactly once and computes the same bits as
    the launch in id order; a table that is no partition of a block (all zero) gives id order; the class decays by one per step."""
    import ctypes as C
    cfg = workloads.s64_config()
    opt = BatchTrajOptimizer(cfg)
    try:
        wl = workloads.generate(cfg, lambda a, ro, r, s: opt.plan(a, ro, r, s, 0, True, True), n_swarms=n_swarms, swarm_size=swarm_size, seed=78)
        db = opt.make_device_batch(wl["agents"], wl["row_offsets"], wl["rows"], None, 0, want_iters=True)
        n = db.n_agents
        db.launch()
        first = db.results()
        m = [db.new_hint_masks() for _ in range(3)]
        m[2].fill_(-1)  # the table the launch is told to clear
        db.set_hint_masks(None, m[0], m[2])
        db.launch()
        torch.cuda.synchronize()
        assert int(m[2].abs().sum()) == 0
        masks = m[0].cpu().numpy().view(np.uint64)
        cls = _classes(first["iters"][:, 0])
        want = np.zeros_like(masks)
        for a in range(n):
            want[a // 64, cls[a]] |= np.uint64(1) << np.uint64(a % 64)
        assert np.array_equal(masks, want)
        # the order the next launch takes from them
        order = torch.full((64 * ((n + 63) // 64),), -7, dtype=torch.int32, device="cuda")
        rc = opt.lib.dlscqp_mask_order_preview_device(opt.handle, C.c_void_p(m[0].data_ptr()), n, C.c_void_p(order.data_ptr()), None)
        assert rc == 0
        torch.cuda.synchronize()
        got = order.cpu().numpy()
        assert np.array_equal(got, _mask_order(masks, n))
        assert np.array_equal(np.sort(got[got >= 0]), np.arange(n))
        # ... plans the same bits, and leaves masks of its own: this step's class or one below the class it was scheduled with
        db.set_hint_masks(m[0], m[1], m[2])
        db.launch()
        second = db.results()
        for k in ("status", "goal", "ctrl", "cost", "iters"):
            assert np.array_equal(first[k], second[k]), k
        masks2 = m[1].cpu().numpy().view(np.uint64)
        assert np.array_equal(masks2, want)  # (the same counts again: max(class, class - 1) = class)
        # a table that is not a partition (all zero) = id order
        z = db.new_hint_masks()
        rc = opt.lib.dlscqp_mask_order_preview_device(opt.handle, C.c_void_p(z.data_ptr()), n, C.c_void_p(order.data_ptr()), None)
        assert rc == 0
        torch.cuda.synchronize()
        assert np.array_equal(order.cpu().numpy(), _mask_order(z.cpu().numpy().view(np.uint64), n))
        # decay: an agent scheduled from class c that needs no iteration now reports class c - 1
        hi = db.new_hint_masks()
        hv = np.zeros((hi.shape[0], 8), np.uint64)
        for a in range(n):
            hv[a // 64, 5] |= np.uint64(1) << np.uint64(a % 64)  # everybody in class 5
        hi.copy_(torch.from_numpy(hv.view(np.int64)).cuda())
        out = db.new_hint_masks()
        db.set_hint_masks(hi, out, None)
        db.launch()
        torch.cuda.synchronize()
        o = out.cpu().numpy().view(np.uint64)
        want3 = np.zeros_like(o)
        for a in range(n):
            want3[a // 64, max(cls[a], 4)] |= np.uint64(1) << np.uint64(a % 64)
        assert np.array_equal(o, want3)
    finally:
        opt.close()
