"""bench.py --gpus N on the one-device multi-process rig: two ranks, BOTH on cuda:0 (the switch
TPP_BENCH_ONE_DEVICE=1: process group on gloo, the peer-store gather over real hipIpcMemHandles exactly as between two GPUs of a
node), launched the way the driver launches it. The line must say: 2 ranks, the strong-scaled MLP as the headline with the gather
inside the timed step, BOTH gather paths timed in the same run, and the gathered output bit-identical to the unsharded result.
(The timings of such a run are meaningless - the ranks time-slice one GPU - and the line says so.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def bench_lines(stdout):
    """bench.py prints TWO JSON lines (round 6): the full detail first, the compact contract line LAST (< 6000 bytes, so the driver's
    8 KB tail always holds it whole). Returns (detail, final)."""
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 2, stdout[-2000:]
    assert stdout.rstrip().endswith(lines[-1]), "the contract line must be the LAST thing on stdout"
    assert len(lines[-1]) < 6000, len(lines[-1])
    detail, final = json.loads(lines[0])["bench_detail"], json.loads(lines[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data"):
        assert final[k] == detail[k], k
    assert final["roofline"]["frac"] == detail["roofline"]["frac"]
    if detail.get("mlp"):
        assert final["roofline"]["configs"]["C4"]["ms_per_step"] == detail["mlp"]["ms_per_step"]
    return detail, final


def test_bench_two_ranks_on_one_device_reports_verified_strong_scaling():
    env = dict(os.environ, TPP_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "3",
           "--no-cpu-baseline", "--no-pmc"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    if r.returncode != 0:
        pytest.fail("bench.py --gpus 2 exited %d\n--- stdout (tail)\n%s\n--- stderr (tail)\n%s" % (r.returncode, r.stdout[-1500:], r.stderr[-6000:]), pytrace=False)
    d, final = bench_lines(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 3
    pgp = d["process_group"]
    assert pgp["world_size"] == 2 and len(pgp["ranks"]) == 2 and pgp["one_device_test_rig"] is True
    # the headline at N > 1: the strong-scaled MLP, gather inside the timed step; C2 (weak, no communication) is secondary
    assert d["scaling"] == "strong" and d["dtype"] == "bf16" and "MLP" in d["metric"]
    assert d["c2_weak"]["scaling"] == "weak" and d["c2_weak"]["dtype"] == "f32" and d["c2_weak"]["value"] > 0
    mlp = d["mlp"]
    assert d["value"] == mlp["value"] and abs(d["ms_per_step"] - mlp["ms_per_step"]) < 1e-9
    # both gather paths were timed in this run and both were checked against the unsharded result
    g = mlp["gathers"]
    assert set(g) == {"peer", "rccl"}, g
    for path in ("peer", "rccl"):
        assert g[path]["ms_per_step"] > 0 and g[path]["gathered_bit_identical"] is True, g
    assert mlp["gathered_bit_identical"] is True and d["config"]["gathered_bit_identical"] is True
    assert mlp["gather"] in ("peer", "rccl") and mlp["one_gpu_same_run"]["ms_per_step"] > 0
    assert mlp["speedup_vs_one_gpu_same_run"] > 0
    assert "TEST RIG" in d["data"]
    assert final["process_group"]["world_size"] == 2 and final["config"]["gathered_bit_identical"] is True
    assert final["roofline"]["configs"]["C4"]["gathered_bit_identical"] is True


def test_bench_gpus_2_without_a_launcher_starts_its_own_ranks():
    """`python3 bench.py --gpus 2 --steps 6 --warmup 3` - NO torchrun, no WORLD_SIZE - must start its own two ranks
    and print ONE line with n_gpus 2 (never an N = 1 line under an N > 1 command)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TPP_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    if r.returncode != 0:
        pytest.fail("bench.py --gpus 2 exited %d\n--- stdout (tail)\n%s\n--- stderr (tail)\n%s" % (r.returncode, r.stdout[-1500:], r.stderr[-6000:]), pytrace=False)
    d, final = bench_lines(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 3
    assert d["process_group"]["world_size"] == 2 and len(d["process_group"]["ranks"]) == 2
    g = d["mlp"]["gathers"]
    assert set(g) == {"peer", "rccl"} and all(g[p]["gathered_bit_identical"] is True for p in g), g
    assert d["mlp"]["gathered_bit_identical"] is True


def test_bench_survives_a_peer_path_that_fails_on_one_rank():
    """a rank whose peer-store gather fails (here: injected on rank 1 behind its timed region) must not cost the line or hang the
    job: every rank votes, all of them drop the path, the RCCL path is the headline and the failure is in the line"""
    env = dict(os.environ, TPP_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", TPP_BENCH_TEST_FAIL_PEER="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "3",
           "--no-cpu-baseline", "--no-pmc"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    if r.returncode != 0:
        pytest.fail("bench.py --gpus 2 exited %d\n--- stdout (tail)\n%s\n--- stderr (tail)\n%s" % (r.returncode, r.stdout[-1500:], r.stderr[-6000:]), pytrace=False)
    mlp = bench_lines(r.stdout)[0]["mlp"]
    assert "failed" in mlp["gathers"]["peer"] and "injected" in mlp["gathers"]["peer"]["failed"] or "another rank" in mlp["gathers"]["peer"]["failed"]
    assert mlp["gather"] == "rccl" and mlp["gathers"]["rccl"]["gathered_bit_identical"] is True and mlp["gathered_bit_identical"] is True
