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
                          "--no-latency", *flags], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields(gpu_required):
    d = _bench()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["unit"] == "QP/s" and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["config"]["agents_per_gpu"] == 512 and d["value"] > 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["qp_per_launch"] == 512
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0


@pytest.mark.parametrize("mode", ["pipelined", "serial"])
def test_multi_gpu_code_path_with_one_rank(gpu_required, mode):
    d = _bench("--force-dist", "--no-cpu-baseline", "--exchange", mode)
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert ("communication stream" in d["config"]["exchange"]) == (mode == "pipelined")
