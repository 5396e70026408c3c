#!/usr/bin/env python3
"""Benchmark of the hot path: batched tube-tracking-MPC QP solves on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W      (N > 1, one rank per GPU)

Workload (BASELINE.json configs[1]): linearised cartpole n=4, m=1, horizon N=10, fixed
initial state (results_linear_system.py:120), batch 4096 (x_k, ref) pairs PER GPU (weak
scaling), the pairs drawn from the closed-loop transients in tests/golden (synthetic, seeded).
One step = one tmpc_solve_batch_device call over the batch, inputs already in HBM.

Metric: QP solves per second (= MPC steps per second), whole job.  The same JSON line
carries
  roofline      FP64 flops of the interior-point iterations actually executed (iters per
                instance come back from the device) over the average kernel duration from HIP
                events on the solve stream, against the MI355X FP64 peak;
  cpu_baseline  the CPU oracle (oracle/tmpc_oracle.c, OpenMP over the batch) timed on this
                box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP64_PEAK_TFLOPS = 78.6      # MI355X FP64 vector = FP64 matrix peak (AMD datasheet)
HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md


def flops_per_iteration(nv, nc):
    """SURVEY.md 8(d) / BASELINE.md 4: form G'DG (symmetric), Cholesky, two solve pairs,
    four G / G' products."""
    return nc * nv * nv + nv ** 3 / 3.0 + 8.0 * nc * nv + 4.0 * nv * nv


def host_cores():
    """Cores this process may actually use: the smaller of its affinity mask and its cgroup CPU quota."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return cores


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="QP instances per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-closed-loop", action="store_true", help="skip the closed-loop extra (profiling runs: keeps the kernel statistics to the timed steps)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist
    import common
    from LinearMPCOverNetworks import _native

    # TMPC_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks share the devices, the
    # statistics gather goes through host memory); the driver's runs use nccl (= RCCL), one rank per GPU
    backend = os.environ.get("TMPC_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    mpc, w = common.make_mpc("cartpole", 10, True, create=True, device=dev_index)
    h = mpc._handle
    nv, nc, npar = _native.get_dims(h, 0)
    nx, nu, N = 4, 1, 10
    B = args.batch

    # synthetic batch: trajectories are sharded, each rank draws its own seeded sample
    S = np.load(os.path.join(common.GOLDEN, "cartpole_N10_states.npy"))
    idx = np.random.default_rng(1000 + rank).integers(0, len(S), B)
    x = torch.from_numpy(S[idx, :nx].copy()).to(dev)
    r = torch.from_numpy(S[idx, nx:].copy()).to(dev)
    u = torch.empty((B, N, nu), dtype=torch.float64, device=dev)
    x0 = torch.empty((B, nx), dtype=torch.float64, device=dev)
    ss = torch.empty((B, nx + nu), dtype=torch.float64, device=dev)
    st = torch.empty(B, dtype=torch.int32, device=dev)
    it = torch.empty(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def step():
        _native.solve_batch_device(h, B, x.data_ptr(), r.data_ptr(), None, u.data_ptr(), x0.data_ptr(), ss.data_ptr(), None,
                                   st.data_ptr(), it.data_ptr())

    def fence():
        _native.synchronize(h)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    from LinearMPCOverNetworks import montecarlo

    def gather_stats():
        # the only exchange of the path: per-trajectory statistics, gathered once per sweep
        # (RCCL all-gather over xGMI; 8 B per trajectory, latency-bound)
        stats = torch.stack([st, it], dim=1).contiguous()
        if backend != "nccl":
            stats = stats.cpu()
        return montecarlo.gather_statistics(stats, world * B, rank, world)

    for _ in range(args.warmup):
        step()
    _native.synchronize(h)
    gather_stats()                      # warm the collective and torch's own kernels as well
    fence()
    _native.kernel_ms_total(h, reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    _native.synchronize(h)
    stats_all = gather_stats()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    kern_ms, launches = _native.kernel_ms_total(h, reset=True)
    stats_np = stats_all.cpu().numpy()
    status_all, iters_all = stats_np[:, 0], stats_np[:, 1]
    iters_local = it.cpu().numpy()

    if rank == 0:
        value = world * B * args.steps / elapsed
        avg_kernel_s = kern_ms / max(launches, 1) * 1e-3
        f_it = flops_per_iteration(nv, nc)
        flops_launch = float(iters_local.sum()) * f_it
        achieved = flops_launch / avg_kernel_s / 1e12
        bytes_solve = 8 * (2 * nx) + 8 * (N * nu + nx + nx + nu) + 8      # inputs + the outputs this call writes + status/iters
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "QP solves/sec (= MPC steps/sec) at batch",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "cartpole n=4 m=1 N=10 tube-tracking QP, fixed x0, batch 4096 per GPU, "
                                   "closed-loop transient (x_k, ref) pairs (BASELINE configs[1])",
                       "batch_per_gpu": B, "nv": nv, "nc": nc, "parallelism": f"trajectory-sharded x{world}",
                       "mean_ipm_iters": float(iters_all.mean()), "optimal_fraction": float((status_all == 0).mean())},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic,
                         "kernel": "tmpc::solve_kernel<12,2,6,7,false,%d>" % (8 if B > 1024 else 4), "avg_kernel_ms": avg_kernel_s * 1e3,
                         "flops_per_launch": flops_launch,
                         "note": "bound is FP64 arithmetic (vector ALU; 78.6 TFLOP/s is also the FP64 MFMA peak), "
                                 "not HBM: algorithmic HBM bytes are %d B/solve" % bytes_solve,
                         "hbm": {"achieved": bytes_solve * B / avg_kernel_s / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                 "frac": bytes_solve * B / avg_kernel_s / 1e9 / HBM_PEAK_GBPS}},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle.oracle import Oracle
            orc = Oracle(mpc._problem_dict())
            cores = host_cores()
            ns = 131072
            ii = np.random.default_rng(5).integers(0, len(S), ns)
            Xc, Rc = S[ii, :nx].copy(), S[ii, nx:].copy()
            orc.solve(Xc[:256], Rc[:256], nthreads=cores)
            tc = time.perf_counter()
            oc = orc.solve(Xc, Rc, nthreads=cores)
            tc = time.perf_counter() - tc
            out["cpu_baseline"] = {"value": ns / tc, "unit": "solves/s", "cores": cores, "kind": "port",
                                   "sample": f"{ns} instances of the same workload, oracle/tmpc_oracle.c (IPM + refinement), "
                                             f"OpenMP over the batch, {tc:.2f} s wall, mean iters {oc['iters'].mean():.2f}"}
            # the reference's Monte-Carlo loop body (results_linear_system.py:209-291) on the host cores: numpy state
            # machines around the same CPU solver, all trajectories of a time step solved together
            def cpu_packets(x_hat, r, gamma=None):
                sol = orc.solve(x_hat, r, gamma, nthreads=cores)
                u_ss = sol["u_ss"] + sol["x_ss"] @ mpc._K.T
                U = np.concatenate([sol["u_nom"], u_ss[:, None, :]], axis=1).transpose(0, 2, 1)
                return np.ascontiguousarray(U), sol["x_nom0"], sol["status"]
            nbc, Tc = 2048, 50
            thc, gac, wdc = montecarlo.draw_realisations(nbc, Tc, w["w_bound"], seed=99)
            tcl = time.perf_counter()
            montecarlo.run_remote_tube_mpc(cpu_packets, w["A"], w["B"], mpc.get_steady_state_controller_gain(),
                                           mpc.get_ancillary_controller_gain(), mpc._N, mpc._Z, np.full(nbc, 0.3),
                                           0.5 * np.ones(Tc), thc, gac, wdc)
            tcl = time.perf_counter() - tcl
            out["cpu_baseline"]["closed_loop"] = {"value": nbc * Tc / tcl, "unit": "MPC steps/s", "trajectories": nbc, "steps": Tc,
                                                  "p_loss": 0.3, "note": "same loop as closed_loop below, solver and state machines on the host"}
        if world == 1 and not args.no_closed_loop:
            # the same kernel inside the device-resident closed loop over the lossy network (tmpc_mc_run): every step is
            # the solve + the estimator / actuator / plant state machines, 4096 trajectories, p_loss = 0.3 (configs[1])
            Tcl = 50
            th, ga, wd = montecarlo.draw_realisations(B, Tcl, w["w_bound"], seed=99)
            pl = np.full(B, 0.3)
            mpc.run_closed_loop(pl[:64], 0.5 * np.ones(Tcl), th[:64], ga[:64], wd[:64])          # warm-up
            tcl = time.perf_counter()
            cl = mpc.run_closed_loop(pl, 0.5 * np.ones(Tcl), th, ga, wd)
            tcl = time.perf_counter() - tcl
            out["closed_loop"] = {"value": B * Tcl / tcl, "unit": "MPC steps/s", "trajectories": B, "steps": Tcl, "p_loss": 0.3,
                                  "tube_violations": int(cl["tube_violations"].sum()), "non_optimal_solves": int(cl["not_optimal"].sum()),
                                  "note": "end to end incl. upload of the realisations and download of the statistics"}
            # offline stage extra: support-function LPs over this workload's terminal set in one launch (tmpc_lp_batch)
            from LinearMPCOverNetworks import _native as nat
            Xf = mpc._Xf
            dirs = np.random.default_rng(7).standard_normal((65536, Xf.A.shape[1]))
            nat.lp_batch(Xf.A, Xf.b, dirs[:256])
            tlp = time.perf_counter()
            lp = nat.lp_batch(Xf.A, Xf.b, dirs)
            tlp = time.perf_counter() - tlp
            out["offline_lp"] = {"value": len(dirs) / tlp, "unit": "LP/s", "rows": int(Xf.A.shape[0]), "dim": int(Xf.A.shape[1]),
                                 "batch": len(dirs), "solved": int((lp["status"] == 0).sum()),
                                 "note": "support LPs over the terminal set, host buffers in and out (set-up stage, DESIGN.md 7a)"}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
