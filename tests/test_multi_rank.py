"""N>1 path on CPU: two gloo ranks shard a sweep, each fills its shard, the statistics
all-gather reproduces the single-process table.  (On the GPUs the same code runs over
nccl = RCCL; bench.py --gpus N uses it.)"""
import os
import subprocess
import sys
import textwrap

import numpy as np

import common
from LinearMPCOverNetworks import montecarlo


def test_shard_bounds_cover_exactly_once():
    for n in (0, 1, 7, 4096, 10000):
        for world in (1, 2, 3, 8):
            b = [montecarlo.shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_trajectory_table_balances_loss_rates():
    pi, si = montecarlo.trajectory_table([0, .1, .2, .3, .4, .5, .6, .7, .8, .9], 1000)
    assert len(pi) == 10000
    lo, hi = montecarlo.shard_bounds(10000, 3, 8)
    counts = np.bincount(pi[lo:hi], minlength=10)
    assert counts.max() - counts.min() <= 1            # every GPU sees every loss rate equally often
    assert set(zip(pi.tolist(), si.tolist())) == {(a, b) for a in range(10) for b in range(1000)}


WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, os.environ["TMPC_PKG"])
    from LinearMPCOverNetworks import montecarlo
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n = 1001                                   # odd on purpose: unequal shards
    lo, hi = montecarlo.shard_bounds(n, rank, world)
    g = torch.arange(lo, hi, dtype=torch.float64)
    local = torch.stack([g, g * g, torch.full_like(g, float(rank))], dim=1)
    table = montecarlo.gather_statistics(local, n, rank, world)
    full = torch.arange(n, dtype=torch.float64)
    assert table.shape == (n, 3)
    assert torch.equal(table[:, 0], full) and torch.equal(table[:, 1], full * full)
    assert int(table[:, 2].sum()) == sum((montecarlo.shard_bounds(n, r, world)[1] - montecarlo.shard_bounds(n, r, world)[0]) * r for r in range(world))
    dist.barrier()
    if rank == 0:
        print("GATHER_OK", world)
    dist.destroy_process_group()
''')


def test_two_rank_gloo_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, TMPC_PKG=common.PKG, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29613", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "GATHER_OK 2" in out.stdout


SWEEP_WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, os.environ["TMPC_TESTS"])
    import common
    from LinearMPCOverNetworks import montecarlo
    from oracle.oracle import Oracle

    class OracleBackedMPC:
        """Stands in for TubeTrackingMPC in this CPU test: same determine_packets contract, solved by the oracle."""
        def __init__(self):
            self.mpc, self.model = common.make_mpc("cartpole", 10, True)
            self.orc = Oracle(self.mpc._problem_dict())
            self._N, self._Z = self.mpc._N, self.mpc._Z
        def get_steady_state_controller_gain(self): return self.mpc.get_steady_state_controller_gain()
        def get_ancillary_controller_gain(self): return self.mpc.get_ancillary_controller_gain()
        def determine_packets(self, x_hat, r, variant=None):
            sol = self.orc.solve(x_hat, r, variant)
            u_ss = sol["u_ss"] + sol["x_ss"] @ self.mpc._K.T
            U = np.concatenate([sol["u_nom"], u_ss[:, None, :]], axis=1).transpose(0, 2, 1)
            return np.ascontiguousarray(U), sol["x_nom0"], sol["status"]

    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    if world > 1:
        dist.init_process_group("gloo")
    m = OracleBackedMPC()
    table, pi = montecarlo.mc_sweep(m, m.model, [0.0, 0.3, 0.6], 3, 12, 0.5, rank=rank, world=world)
    if rank == 0:
        np.save(os.environ["TMPC_OUT"], table)
        print("SWEEP_OK", world, table.shape)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
''')


def test_sweep_is_independent_of_the_shard_count(tmp_path, oracle_lib):
    """BASELINE config 4 in miniature: 3 loss rates x 3 seeds x 12 steps, closed loop per trajectory; the
    gathered statistics of a 2-rank run (gloo) equal the single-process table bit for bit."""
    script = tmp_path / "sweep_worker.py"
    script.write_text(SWEEP_WORKER)
    tables = {}
    for world in (1, 2):
        out_path = tmp_path / f"table_{world}.npy"
        env = dict(os.environ, TMPC_TESTS=os.path.join(common.ROOT, "tests"), TMPC_OUT=str(out_path), MASTER_ADDR="127.0.0.1",
                   OMP_NUM_THREADS="2")
        if world == 1:
            cmd = [sys.executable, str(script)]
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                   "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)]
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        assert f"SWEEP_OK {world}" in out.stdout
        tables[world] = np.load(out_path)
    assert tables[1].shape == (9, 3)
    assert np.array_equal(tables[1], tables[2])
    assert np.all(tables[1][:, 1] == 0) and np.all(tables[1][:, 2] == 0)       # no tube violation, all solves optimal
