"""-m gpu: the library's own exchange of the solved plans (dlscqp_exchange_*: ncclAllGather called from C on a stream the exchange
owns) and one call per replanning step (dlscqp_step_device) -- with ONE rank, which is what a one-GPU box can run: a one-rank RCCL
communicator (real ncclAllGather) and the local_only variant (no librccl).  The two-rank logic is covered on the CPU tier over gloo
(tests/test_dist.py, PlanExchange: same record layout, same post / before_overwrite / table protocol)."""
import ctypes as C

import numpy as np
import pytest
import torch

from dlsc_gc_planner_amd import abi, sharding, workloads
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer

pytestmark = pytest.mark.gpu


def _batches(opt, cfg, n_swarms, seed, parts):
    wl = workloads.generate(cfg, lambda a, ro, r, s: opt.plan(a, ro, r, s, 0, True, True), n_swarms=n_swarms, seed=seed)
    N, K = len(wl["agents"]), wl["K"]
    per = N // parts
    dbs = [opt.make_device_batch(wl["agents"][g * per:(g + 1) * per], np.arange(per + 1) * K, wl["rows"][g * per * K:(g + 1) * per * K],
                                 None, 0, want_iters=True) for g in range(parts)]
    return wl, dbs, N, per


@pytest.mark.parametrize("local_only", [False, True])
def test_posted_plans_come_back_as_the_gathered_table(gpu_required, local_only):
    cfg = workloads.s64_config()
    opt = BatchTrajOptimizer(cfg)
    try:
        wl, (db,), N, _ = _batches(opt, cfg, 4, 91, 1)
        x = sharding.NativeExchange(opt, 0, 1, N, None, local_only=local_only)
        E = cfg.M * (cfg.n + 1) * 3
        side = torch.cuda.Stream()
        for rep in range(5):  # more posts than buffer pairs: exchange k waits for exchange k - 2
            db.out_ctrl.zero_()
            db.out_status.fill_(-1)
            x.before_overwrite()
            db.launch()
            x.post(db.out_ctrl, db.out_goal, db.out_status)
            table = x.table(side)  # a consumer on another stream
            with torch.cuda.stream(side):
                got = table.clone()
            side.synchronize()
            ctrl, goal, status = x.unpack(got, (torch.float32, torch.float32, torch.int32))
            r = db.results()
            assert got.shape == (N, E + 4)
            assert np.array_equal(ctrl.cpu().numpy(), r["ctrl"].reshape(N, E))
            assert np.array_equal(goal.cpu().numpy(), r["goal"])
            assert np.array_equal(status.cpu().numpy().ravel(), r["status"]) and (r["status"] >= 0).all()
        x.enable_timing(True, 8)
        db.launch()
        x.post(db.out_ctrl, db.out_goal, db.out_status)
        t = x.times()
        assert len(t["pack"]) == 1 and t["pack"][0] > 0 and t["all_gather"][0] > 0 and t["lag"][0] >= t["all_gather"][0]
        x.close()
    finally:
        opt.close()


def test_exchange_rejects_bad_arguments(gpu_required):
    opt = BatchTrajOptimizer(workloads.s64_config())
    try:
        h = C.c_void_p()
        c = abi.ExchangeConfig()
        c.rank, c.world, c.n_local = 0, 2, 64  # two ranks without an id
        assert opt.lib.dlscqp_exchange_create(opt.handle, C.byref(c), C.byref(h)) == abi.ERR_INVALID_ARGUMENT
        c.world, c.rank = 1, 1
        assert opt.lib.dlscqp_exchange_create(opt.handle, C.byref(c), C.byref(h)) == abi.ERR_INVALID_ARGUMENT
        c.rank, c.n_local = 0, 0
        assert opt.lib.dlscqp_exchange_create(opt.handle, C.byref(c), C.byref(h)) == abi.ERR_INVALID_ARGUMENT
        c.world, c.n_local, c.local_only = 2, 64, 1
        assert opt.lib.dlscqp_exchange_create(opt.handle, C.byref(c), C.byref(h)) == abi.ERR_INVALID_ARGUMENT
        assert opt.lib.dlscqp_step_device(None) == abi.ERR_INVALID_ARGUMENT
        uid = sharding.exchange_unique_id(opt.lib)
        assert len(uid) == abi.EXCHANGE_ID_BYTES and any(uid)
    finally:
        opt.close()


@pytest.mark.parametrize("mode", [abi.EXCHANGE_NONE, abi.EXCHANGE_PIPELINED, abi.EXCHANGE_SERIAL])
def test_one_call_plans_every_group_and_posts_the_step(gpu_required, mode):
    """dlscqp_step_device: two groups on two streams (each with its own context), their batches writing slices of one output table,
    the kernel's own launch order from the previous step's counts -- equal to launching the groups one by one; with an exchange the
    gathered table of every step is that step's output table."""
    cfg = workloads.s64_config()
    opts = [BatchTrajOptimizer(cfg) for _ in range(2)]
    try:
        wl, dbs, N, per = _batches(opts[0], cfg, 6, 92, 2)
        dbs[1] = opts[1].make_device_batch(wl["agents"][per:], np.arange(per + 1) * wl["K"], wl["rows"][per * wl["K"]:], None, 0, want_iters=True)
        ref = []
        for db in dbs:
            db.use_row_pitch()
            db.launch()
            ref.append(db.results())
        B = 3  # a rotation of three output tables, as the pipelined exchange asks for
        shape1 = tuple(dbs[0].out_ctrl.shape[1:])
        outs = [(torch.zeros((N,) + shape1, device="cuda"), torch.zeros((N, 3), device="cuda"), torch.full((N,), -1, dtype=torch.int32, device="cuda"))
                for _ in range(B)]
        prev = [torch.from_numpy(r["iters"]).cuda() for r in ref]
        streams = [torch.cuda.Stream() for _ in range(2)]
        x = sharding.NativeExchange(opts[0], 0, 1, N, None) if mode != abi.EXCHANGE_NONE else None
        for step in range(7):
            j = step % B
            groups = (abi.StepGroup * 2)()
            for g, db in enumerate(dbs):
                c_, g_, s_ = outs[j]
                db.use_outputs(c_[g * per:(g + 1) * per], g_[g * per:(g + 1) * per], s_[g * per:(g + 1) * per])
                db.set_hint([prev[g]])
                db.c.do_goal, db.c.do_traj, db.c.launch_offset, db.c.launch_count = 1, 1, 0, 0
                groups[g].ctx, groups[g].batch, groups[g].stream = opts[g].handle, C.pointer(db.c), streams[g].cuda_stream
            a = abi.StepArgs()
            a.n_groups, a.groups, a.exchange_mode = 2, groups, mode
            if x is not None:
                a.exchange = x.handle
                a.ctrl, a.goal, a.status = (t.data_ptr() for t in outs[j])
                x.last, x.k = x.k & 1, x.k + 1
            torch.cuda.synchronize()
            outs[j][0].zero_()
            outs[j][2].fill_(-1)
            torch.cuda.synchronize()
            assert opts[0].lib.dlscqp_step_device(a) == abi.SUCCESS, opts[0].lib.dlscqp_last_error()
            torch.cuda.synchronize()
            want_ctrl = np.concatenate([r["ctrl"] for r in ref])
            want_status = np.concatenate([r["status"] for r in ref])
            assert np.array_equal(outs[j][0].cpu().numpy(), want_ctrl) and np.array_equal(outs[j][2].cpu().numpy(), want_status)
            if x is not None:
                ctrl, goal, status = x.unpack(x.table(), (torch.float32, torch.float32, torch.int32))
                torch.cuda.synchronize()
                assert np.array_equal(ctrl.cpu().numpy(), want_ctrl.reshape(N, -1))
                assert np.array_equal(status.cpu().numpy().ravel(), want_status)
        if x is not None:
            x.close()
    finally:
        for o in opts:
            o.close()


def test_plans_written_as_records_are_sent_as_they_lie(gpu_required):
    """dlscqp_device_batch.out_record_stride + dlscqp_exchange_post_records: the plan kernel writes control points | sub-goal |
    verdict per agent into the table the exchange sends -- no packing pass; the gathered table equals the dense outputs of the same
    batch, through dlscqp_step_device as well."""
    cfg = workloads.s64_config()
    opt = BatchTrajOptimizer(cfg)
    try:
        wl, (db,), N, _ = _batches(opt, cfg, 5, 93, 1)
        db.launch()
        ref = db.results()
        E = cfg.M * (cfg.n + 1) * 3
        rec = torch.zeros((N, E + 4), device="cuda")
        rec[:, E + 3:].view(torch.int32).fill_(-1)
        db.use_records(rec)
        db.use_row_pitch()
        db.set_hint([torch.from_numpy(ref["iters"]).cuda()])
        x = sharding.NativeExchange(opt, 0, 1, N, None)
        db.launch()
        got = db.results()
        for k in ("ctrl", "goal", "status", "cost", "goal_status"):
            assert np.array_equal(got[k], ref[k]), k
        x.post_records(rec)
        ctrl, goal, status = x.unpack(x.table(), (torch.float32, torch.float32, torch.int32))
        torch.cuda.synchronize()
        assert np.array_equal(ctrl.cpu().numpy(), ref["ctrl"].reshape(N, E)) and np.array_equal(goal.cpu().numpy(), ref["goal"])
        assert np.array_equal(status.cpu().numpy().ravel(), ref["status"])
        # ... and as one call
        rec.zero_()
        groups = (abi.StepGroup * 1)()
        db.c.do_goal, db.c.do_traj, db.c.launch_offset, db.c.launch_count = 1, 1, 0, 0
        groups[0].ctx, groups[0].batch, groups[0].stream = opt.handle, C.pointer(db.c), torch.cuda.current_stream().cuda_stream
        a = abi.StepArgs()
        a.n_groups, a.groups, a.exchange_mode, a.exchange, a.records = 1, groups, abi.EXCHANGE_PIPELINED, x.handle, rec.data_ptr()
        for _ in range(3):
            assert opt.lib.dlscqp_step_device(a) == abi.SUCCESS, opt.lib.dlscqp_last_error()
            x.last, x.k = x.k & 1, x.k + 1
        t = x.table().clone()
        torch.cuda.synchronize()
        assert torch.equal(t, rec)
        assert np.array_equal(t[:, :E].cpu().numpy(), ref["ctrl"].reshape(N, E))
        # going back to dense outputs switches the record stride off
        db.use_outputs(torch.zeros_like(torch.from_numpy(ref["ctrl"])).cuda(), torch.zeros((N, 3), device="cuda"), torch.full((N,), -1, dtype=torch.int32, device="cuda"))
        db.launch()
        again = db.results()
        assert np.array_equal(again["ctrl"], ref["ctrl"]) and np.array_equal(again["status"], ref["status"])
        x.close()
    finally:
        opt.close()


def test_a_post_that_does_not_pack_on_the_communication_stream_may_be_overwritten_at_once(gpu_required):
    """NativeExchange.post(pack_on_comm=False): the caller rewrites its three tensors on its stream right after the post (bench.py's
    sharded mission does: `own_ctrl.index_copy_` of the next step); the record that is gathered must be the one that was posted.  The
    communication stream is held back by a long copy queued in front of the post, so that without the wait the packing would run
    long after the tensors have been overwritten."""
    cfg = workloads.s64_config()
    opt = BatchTrajOptimizer(cfg)
    try:
        N, E = 4096, cfg.M * (cfg.n + 1) * 3
        x = sharding.NativeExchange(opt, 0, 1, N, None, local_only=True)
        big = torch.zeros(64 << 20, dtype=torch.float32, device="cuda")
        for rep in range(6):
            ctrl = torch.full((N, E), float(rep + 1), dtype=torch.float32, device="cuda")
            goal = torch.full((N, 3), float(rep + 1) * 0.5, dtype=torch.float32, device="cuda")
            status = torch.full((N,), rep, dtype=torch.int32, device="cuda")
            # two exchanges in flight occupy the communication stream (each waits for its own ready event: give them slow ones)
            for _ in range(4):
                big.add_(1.0)
            x.post(ctrl, goal, status, pack_on_comm=False)
            ctrl.fill_(-1.0)
            goal.fill_(-1.0)
            status.fill_(-9)
            got = x.table().clone()
            torch.cuda.synchronize()
            c, g, s = x.unpack(got, (torch.float32, torch.float32, torch.int32))
            assert float(c.min()) == float(c.max()) == float(rep + 1)
            assert float(g.min()) == float(g.max()) == float(rep + 1) * 0.5
            assert int(s.min()) == int(s.max()) == rep
        x.close()
    finally:
        opt.close()


@pytest.mark.parametrize("tables", [3, 4, 8])
def test_a_rotation_of_tables_waits_for_the_post_that_read_the_table(gpu_required, tables):
    """dlscqp_exchange_before_overwrite_rotating: a caller that rotates through B record tables rewrites table k mod B at step k once
    post k - B has read it -- not earlier (the gathered tables are the posted ones although every table is rewritten as soon as the
    wait lets the stream go, with the posts held back by slow producers) and without waiting for the post before last (the event it
    waits for is the one of post k - B: checked through the argument range and the first B steps, which wait for nothing)."""
    cfg = workloads.s64_config()
    opt = BatchTrajOptimizer(cfg)
    try:
        N, W = 8192, cfg.M * (cfg.n + 1) * 3 + 4
        x = sharding.NativeExchange(opt, 0, 1, N, None, local_only=True)
        assert opt.lib.dlscqp_exchange_before_overwrite_rotating(x.handle, None, 1) == abi.ERR_INVALID_ARGUMENT
        assert opt.lib.dlscqp_exchange_before_overwrite_rotating(x.handle, None, 9) == abi.ERR_INVALID_ARGUMENT
        rec = [torch.zeros((N, W), device="cuda") for _ in range(tables)]
        big = torch.zeros(32 << 20, dtype=torch.float32, device="cuda")
        st, side = torch.cuda.Stream(), torch.cuda.Stream()
        got, cloned = [], None
        for step in range(3 * tables + 2):
            j = step % tables
            with torch.cuda.stream(st):
                x.before_overwrite(st, tables=tables)
                rec[j].fill_(float(step + 1))  # (what the plan kernel of step `step` writes)
                for _ in range(2):
                    big.add_(1.0)              # a slow producer: the post's ready event comes late, the next steps' writes are queued behind
            if cloned is not None:
                st.wait_event(cloned)  # (a gathered table is valid until the post after next: the consumer has copied the previous one)
            x.post_records(rec[j], stream=st)
            t = x.table(side)
            with torch.cuda.stream(side):
                got.append(t.clone())
                cloned = torch.cuda.Event()
                cloned.record(side)
        torch.cuda.synchronize()
        for step, g in enumerate(got):
            assert float(g.min()) == float(g.max()) == float(step + 1), (step, float(g.min()), float(g.max()))
        x.close()
    finally:
        opt.close()
