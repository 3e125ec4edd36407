"""bench.py end to end on a small workload: the one-GPU line carries the contract's fields, and the N > 1 control flow (eager leg,
guarded graph leg, one JSON line from rank 0) runs with two ranks sharing the box's one GPU over gloo (RNAD_BENCH_REHEARSAL)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
SMALL = ["--depth", "4", "--batch-log2", "14", "--steps", "40", "--warmup", "3", "--other-steps", "4", "--cpu-lanes-log2", "9"]


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line expected:\n{out[-3000:]}"
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_one_gpu_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL], capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_line(r.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 40 and j["warmup"] == 3 and j["value"] > 0 and j["unit"] == "env-steps/s"
    assert j["net_evaluation"]["step_replayed_from_hipGraph"] is True
    roof = j["roofline"]
    assert roof["bound"] in ("hbm", "mfma", "valu", "l1/latency") and roof["unit"] in ("GB/s", "TFLOP/s", "Gwave-inst/s") and 0 < roof["frac"] < 1
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    cpu = j["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["sample"]
    assert abs(j["value"] - j["config"]["global_batch"] * j["config"]["T"] * j["steps"] / (j["ms_per_step"] * 1e-3 * j["steps"])) < 1e-6 * j["value"]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("scaling", ("weak", "strong"))
def test_two_rank_control_flow(scaling):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RNAD_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scaling", scaling, *SMALL]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=850, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["scaling"] == scaling and j["config"]["global_batch"] == 2 * j["config"]["per_gpu_batch"]
    assert j["config"]["global_batch"] == ((2 << 14) if scaling == "weak" else (1 << 14))
    assert j["value"] > 0 and "eager" in j["legs_ms_per_step"] and "cpu_baseline" not in j
    assert j["collectives"]["ranks"] == 2 and j["collectives"]["backend"] == "gloo" and j["collectives"]["rccl_ranks"] is None
    if scaling == "strong":  # rank 0 also timed the same global batch alone: the N = 1 point of the curve
        base = j["strong_scaling"]["base"]
        assert base["global_batch"] == 1 << 14 and base["ms_per_step"] > 0 and j["strong_scaling"]["speedup_vs_one_gpu_same_batch"] > 0
    else:
        assert "strong_scaling" not in j
    assert abs(j["value"] - j["config"]["global_batch"] * j["config"]["T"] / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]


@pytest.mark.timeout(1500)
def test_eight_ranks_with_no_other_flags_run_configs_2():
    """`bench.py --gpus 8` as the driver launches it: BASELINE.json configs[2] -- one 2^22 batch, 2^19 lanes per rank, strong scaling --
    rehearsed with the eight ranks sharing this box's GPU over gloo.  Rank 0's stand-alone leg is the 2^22 batch on ONE GPU in the
    default mode (the N = 1 point of that curve)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RNAD_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "2", "--other-steps", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1400, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = _json_line(r.stdout)
    cfg = j["config"]
    assert j["n_gpus"] == 8 and j["scaling"] == "strong" and "configs[2]" in cfg["workload"]
    assert cfg["global_batch"] == 1 << 22 and cfg["per_gpu_batch"] == 1 << 19 and cfg["T"] == 12
    assert j["net_evaluation"]["in_effect"] == "True" and j["collectives"]["ranks"] == 8
    base = j["strong_scaling"]["base"]
    assert base["global_batch"] == 1 << 22 and base["net_mode_in_effect"] == "True" and base["step_replayed_from_hipGraph"] is True
    assert base["ms_per_step"] > 0 and j["value"] > 0
    assert abs(j["value"] - cfg["global_batch"] * cfg["T"] / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]


@pytest.mark.timeout(1500)
def test_eight_ranks_with_row_sharding():
    """The same rehearsal with `--shard-rows`: every rank evaluates 1/8 of the 2S rows, all-gathers the record tables and all-reduces the
    per-row sums (gloo here; the RCCL path differs only in the in-place all_gather_into_tensor)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RNAD_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "10", "--warmup", "2", "--other-steps", "2",
           "--shard-rows", "--no-base-leg"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1400, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 8 and j["collectives"]["shard_rows"] is True and "all_gather" in j["collectives"]["per_step"]
    assert j["net_evaluation"]["in_effect"] == "True" and j["value"] > 0
