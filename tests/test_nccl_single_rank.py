"""The RCCL code path on the one GPU a test box has: a world-size-1 `nccl` process group, started in a fresh subprocess
before anything touches the GPU (exactly as `bench.py` and `scripts/mc_linear_system.py` start theirs under
torch.distributed.run), carries the statistics gather on device tensors, the Monte-Carlo sweep of BASELINE configs[3] at its
full size (10 loss rates x 1000 runs x 250 steps, results_linear_system.py:147-149) and the bench's timed loop.  What it
proves in one process: RCCL initialises next to this library's HIP runtime (the shared-runtime binding of _native.py),
`device_id=` binding works, and the collective accepts the tensors the product hands it."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

import common

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(common.PKG)

WORKER = textwrap.dedent('''
    import os, sys
    import torch, torch.distributed as dist
    sys.path.insert(0, os.environ["TMPC_PKG"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)      # as bench.py does for N > 1
    from LinearMPCOverNetworks import montecarlo, workloads
    n = 1001
    g = torch.arange(n, dtype=torch.float64, device=dev)
    local = torch.stack([g, g * g], dim=1)
    table = montecarlo.gather_statistics(local, n, 0, 1, force_collective=True)
    assert table.is_cuda and torch.equal(table, local)
    # a sweep through the product path: offline sets via the LP kernel, device-resident closed loop, device-side gather
    import numpy as np
    mpc, model = workloads.make_controller("cartpole", 10, True, device=0)
    tab, pi = montecarlo.mc_sweep(mpc, model, np.array([0.0, 0.5]), 8, 40, 0.5, rank=0, world=1, device=dev, on_device=True,
                                  device_rng=True, force_collective=True)
    assert tab.shape == (16, 3) and np.all(np.isfinite(tab[:, 0])) and np.all(tab[:, 1] == 0)
    dist.barrier()
    dist.destroy_process_group()
    print("NCCL_SINGLE_RANK_OK")
''')


def _env(port):
    return dict(os.environ, TMPC_PKG=common.PKG, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")


def test_nccl_group_gather_and_sweep_on_device(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    res = subprocess.run([sys.executable, str(script)], env=_env(29541), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "NCCL_SINGLE_RANK_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]


def test_config4_full_size_through_the_script_with_a_process_group():
    """BASELINE configs[3] at the reference's size, one shard = the whole sweep: 10 x 1000 trajectories x 250 steps, N = 20."""
    env = dict(_env(29542), TMPC_FORCE_PG="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "mc_linear_system.py"), "--n-mc", "1000", "--warm-start",
                          "--device-rng"], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = res.stdout.splitlines()
    head = [l for l in lines if l.startswith("== tube MPC: 10000 trajectories x 250 steps")]
    assert head, res.stdout[-2000:]
    rows = [l.split() for l in lines if l[:1].isspace() or l[:1].isdigit()]
    rows = [r for r in rows if len(r) == 5]
    assert len(rows) == 10
    assert all(int(r[2]) == 0 for r in rows), rows                    # no tube violations (Proposition 2 of the paper)
    assert all(int(r[3]) == 0 and int(r[4]) == 0 for r in rows), rows  # every solve optimal
    te = [float(r[1]) for r in rows]
    assert all(0.0 < t < 0.2 for t in te) and te[-1] > te[0]          # tracking degrades with the loss rate


def test_bench_timed_loop_with_a_process_group():
    env = dict(_env(29543), TMPC_BENCH_FORCE_PG="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--no-cpu-baseline",
                          "--no-closed-loop", "--no-extras"], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 1e6 and line["config"]["optimal_fraction"] == 1.0
    # the line proves by itself that the collective joined the ranks it claims (here: one) and what each of them measured
    assert line["ranks_seen"] == 1
    pr = line["per_rank_ms"]
    assert pr["backend"] == "nccl" and pr["device_index"] == [0]
    assert len(pr["ms_per_step"]) == 1 and len(pr["avg_kernel_ms"]) == 1
    assert 0.0 < pr["avg_kernel_ms"][0] <= pr["ms_per_step"][0] <= line["ms_per_step"] * 1.0001
    assert line["preload"] == "none"


def test_bench_refuses_more_nccl_ranks_than_gpus():
    """Two ranks announced on a one-GPU box: a clear message and exit code 3 before any process group is set up."""
    env = dict(_env(29544), WORLD_SIZE="2", LOCAL_WORLD_SIZE="2")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs")
    assert res.returncode == 3 and "RCCL needs one GPU per rank" in res.stderr, res.stderr[-2000:]


def test_bench_two_ranks_rehearsal_on_one_gpu():
    """The N > 1 path of bench.py with TWO ranks, as the driver launches it (torch.distributed.run, one process per rank), on the
    one GPU a test box has: TMPC_BENCH_BACKEND=gloo lets the ranks share the device (RCCL wants a GPU per rank; the nccl branch is
    covered by the single-rank tests above).  What it checks is the part the driver's scaling record depends on: both ranks join,
    the line says so (`ranks_seen`, `per_rank_ms` with two entries), the value is the whole job's."""
    env = dict(os.environ, TMPC_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                          "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]                      # rank 0 prints the one line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["scaling"] == "weak"
    pr = line["per_rank_ms"]
    assert pr["backend"] == "gloo" and len(pr["ms_per_step"]) == 2 and len(pr["avg_kernel_ms"]) == 2
    assert abs(line["ms_per_step"] - max(pr["ms_per_step"])) < 0.05 * line["ms_per_step"] + 0.05      # the slowest rank's region (up to the barrier)
    assert line["config"]["batch_per_gpu"] == 4096 and line["config"]["optimal_fraction"] == 1.0
    assert abs(line["value"] - 2 * 4096 * 6 / (line["ms_per_step"] * 6e-3)) < 1e-6 * line["value"]        # whole-job solves per second
