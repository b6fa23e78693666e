"""-m gpu: bench.py end to end on a small batch -- the single-GPU line and the N > 1 code path (process group, plan
exchange on the communication stream, double-buffered outputs) exercised with a one-rank RCCL process group, in both
exchange modes.  The 8-GPU run itself is the driver's; this keeps its code path from rotting."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--swarms", "8", "--steps", "6", "--warmup", "2",
                          *flags], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    assert out.stdout.strip() == lines[0], "stdout carries the JSON line and nothing else"
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields(gpu_required):
    d = _bench("--no-extras")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["unit"] == "QP/s" and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["config"]["agents_per_gpu"] == 512 and d["value"] > 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["limiter"] and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # the eight swarms in four groups, each group on its own stream: four launches of 128 agents side by side
    assert r["qp_per_launch"] == 128 and r["launches_per_step"] == 4 and d["config"]["swarm_groups"] == 4
    # the line's roofline figure is the STEP's: all agents' algorithmic bytes over ms_per_step; the per-launch figure and the
    # "if the four launches overlapped fully" figure are named extras
    assert abs(r["achieved"] - 512 * r["algorithmic_bytes_per_qp"] / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-9 * r["achieved"]
    assert abs(r["achieved_if_launches_overlapped_fully"] - 4 * r["achieved_one_launch"]) < 1e-9 * r["achieved_one_launch"] * 4
    assert 0 < r["effective_concurrency"] <= 4.0 + 1e-9
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["port_kind"] == "native scalar solver" and c["cores"] >= 1 and c["value"] > 0 and c["oracle_qp_per_s_one_core"] > 0
    assert c["value"] == c["native_qp_per_s"] and c["lane_serial_qp_per_s"] > 0  # the CPU-shaped solver is the baseline, the lane-serial port beside it
    assert c["value"] > 3 * c["oracle_qp_per_s_one_core"]  # the kernel's algorithm, not the dense checker
    assert d["config"]["distinct_batches"] == 4 and d["config"]["ms_per_step_no_order_hint"] > 0 and d["config"]["ms_per_step_host_order"] > 0
    assert d["config"]["ms_per_step_sorted_on_device"] > 0  # round 3's sort kernels, timed beside the kernel's own order


@pytest.mark.parametrize("mode", ["pipelined", "serial"])
def test_multi_gpu_code_path_with_one_rank(gpu_required, mode):
    d = _bench("--force-dist", "--no-cpu-baseline", "--exchange", mode)
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert ("communication stream" in d["config"]["exchange"]) == (mode == "pipelined")
    # with the exchange the swarms are planned in two groups; ONE all-gather per step carries both groups' plans
    assert d["config"]["swarm_groups"] == 2 and d["roofline"]["launches_per_step"] == 2 and d["roofline"]["qp_per_launch"] == 256
    assert d["config"]["host_ms_to_queue_a_step"] > 0
    e = d["config"]["exchange_modes_ms_per_step"]
    assert e["pipelined"] > 0 and e["serial"] > 0  # both modes are timed in every N > 1 run
    x = d["x2048_strong"]  # BASELINE.json configs[4] through the sharded step (one rank: no halo)
    assert x["agents_total"] == 2048 and x["halo_agents_rank0"] == 0 and x["ms_per_step_serial_exchange"] > 0 and x["solved_fraction_rank0"] > 0.9
    assert x["ms_per_step_pipelined_exchange"] > 0
    # every field a real multi-GPU run will be read by is there, non-null, with a one-rank group already
    assert d["n_gpus_requested"] == 1 and d["n_gpus_visible"] >= 1
    stages = ("plan_kernel_ms", "pack_ms", "all_gather_ms", "output_ready_to_gathered_ms", "wait_for_table_ms")
    for k in stages:
        v = d["per_rank_ms"][k]
        assert v["min"] is not None and v["min"] == v["min"] and v["min"] <= v["mean"] <= v["max"], (k, v)
    assert d["per_rank_ms"]["plan_kernel_ms"]["mean"] > 0 and d["per_rank_ms"]["all_gather_ms"]["mean"] > 0
    for m2 in ("serial", "pipelined"):
        for k in stages[1:]:
            assert x["exchange_per_rank_ms"][m2][k]["mean"] >= 0, (m2, k)
    e60 = d["empty60_sharded"]  # BASELINE.json configs[2]: the mission sharded over the ranks, exchange on the critical path
    assert e60["agents"] == 60 and e60["ranks"] == 1 and e60["ms_per_step"] > 0 and e60["mean_distance_flown_m"] > 0.3
    assert e60["qp_failsafes_last_step_all_ranks"] == 0
    for k in stages:
        assert e60["per_rank_ms"][k]["mean"] > 0, k  # the wait for the table is real here: every agent is a halo agent


def test_exchange_is_checked_whole_and_falls_back(gpu_required, monkeypatch):
    """the first multi-rank run must not rest on the least-tested code.  With N > 1 (here: a one-rank group) the whole
    gathered table is compared with torch.distributed's own all-gather of the same records before and after the timed loop; a table
    that differs in ONE word (test hook) makes the run fall back to torch.distributed, say so, and finish with exit code 0."""
    d = _bench("--force-dist", "--no-cpu-baseline", "--headline-only")
    assert d["config"]["exchange_check"] == {"before": "ok", "after": "ok", "fallback": None}
    assert d["config"]["exchange_issued_by"].startswith("the library")
    assert d["scaling_model"]["n_gpus"] == [1, 2, 4, 8] and all(v > 0 for v in d["scaling_model"]["ms_per_step"])
    monkeypatch.setenv("DLSCQP_BENCH_CORRUPT_EXCHANGE", "1")
    d = _bench("--force-dist", "--no-cpu-baseline", "--headline-only")
    c = d["config"]["exchange_check"]
    assert "1 of" in c["before"] and "differ" in c["before"] and c["fallback"] == c["before"] and c["after"] == "ok"
    assert "FALLBACK" in d["config"]["exchange_issued_by"] and d["value"] > 0


def test_gpus_flag_starts_the_ranks_itself(gpu_required):
    """`bench.py --gpus 2` without RANK in the environment must not silently measure one GPU: it starts the ranks as child
    processes (torch.distributed.run).  On a one-GPU box the second rank cannot get a device, so the run fails -- loudly."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--swarms", "2", "--steps", "4", "--warmup", "1",
                          "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    import torch
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if torch.cuda.device_count() >= 2:
        assert out.returncode == 0 and json.loads(lines[0])["n_gpus"] == 2
    else:
        assert out.returncode != 0 and not lines, (out.returncode, out.stdout[-500:])
        assert "torch.distributed" in out.stderr or "invalid device" in out.stderr.lower() or "ChildFailedError" in out.stderr


def test_two_ranks_end_to_end_on_one_gpu_over_gloo(gpu_required):
    """The multi-rank logic of bench.py with TWO real ranks: RCCL refuses two ranks on one device, gloo moves device tensors
    through the host, so `DLSCQP_BENCH_GLOO_ONE_DEVICE=1` (a test hook of bench.py) runs both ranks on cuda:0 -- shards of the S64
    swarms in two groups per rank with one exchange per step, the check that every gathered table holds the rank's own
    block, X2048 with real halo agents on both ranks (interior launch, exchange, halo launch), empty60 sharded 30 + 30 with the
    exchange on the critical path, the per-rank breakdown gathered over both ranks.  Logic, not timing."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(DLSCQP_BENCH_GLOO_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29737",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--swarms", "8", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_gpus_requested"] == 2 and "TEST HOOK" in d["data"]
    assert d["value"] > 0 and d["config"]["swarm_groups"] == 2 and d["config"]["agents_per_gpu"] == 512
    for k, v in d["per_rank_ms"].items():
        assert v["min"] <= v["mean"] <= v["max"], k
    x = d["x2048_strong"]
    assert x["agents_total"] == 2048 and x["agents_this_rank"] == 1024 and x["halo_agents_rank0"] > 0 and x["interior_agents_rank0"] > 0
    assert x["ms_per_step_serial_exchange"] > 0 and x["ms_per_step_pipelined_exchange"] > 0 and x["solved_fraction_rank0"] > 0.9
    e60 = d["empty60_sharded"]
    assert e60["ranks"] == 2 and e60["agents"] == 60 and e60["qp_failsafes_last_step_all_ranks"] == 0 and e60["mean_distance_flown_m"] > 0.3
