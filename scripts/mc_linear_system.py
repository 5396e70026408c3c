"""Monte-Carlo sweep over packet-loss rates on the GPU(s): counterpart of the reference's
Results/results_linear_system.py (tube MPC part) and ..._with_extendedMPC.py.

    python scripts/mc_linear_system.py --n-mc 20 --T 250 --N 20 [--extended]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/mc_linear_system.py ...

One process per GPU; the sweep is sharded over the ranks, the statistics table is all-gathered (RCCL)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "robust-tracking-mpc-over-lossy-networks_amd"))
from LinearMPCOverNetworks import montecarlo, workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-mc", type=int, default=20)            # results_linear_system.py:147
    ap.add_argument("--T", type=int, default=250)              # :143
    ap.add_argument("--N", type=int, default=20)               # :64
    ap.add_argument("--ref", type=float, default=0.5)          # :160
    ap.add_argument("--extended", action="store_true")
    ap.add_argument("--rmpc", action="store_true", help="the non-robust comparator TrackingMPC (results_linear_system.py:132-140, 262-287) "
                                                        "instead of the tube MPC; reports the infeasible runs per loss rate (:268-270)")
    ap.add_argument("--warm-start", action="store_true", help="tmpc_mc_set_warm_start: previous working set first")
    ap.add_argument("--timing", action="store_true", help="per-solve device times (tmpc_set_solve_timing): the max / quantiles / median "
                                                          "results_linear_system.py:305-315 prints")
    ap.add_argument("--device-rng", action="store_true", help="draw the realisations on the device (tmpc_mc_set_device_rng, Philox keyed "
                                                              "by the global trajectory index) instead of uploading them")
    ap.add_argument("--host-loop", action="store_true", help="state machines in numpy on the host instead of on the device")
    ap.add_argument("--all-controllers", action="store_true",
                    help="tube MPC, extended tube MPC and the tracking MPC one after the other on the SAME realisations "
                         "(results_linear_system_with_extendedMPC.py:267-291 runs the three side by side in one loop)")
    ap.add_argument("--reference-streams", action="store_true",
                    help="replay the reference's own random streams (seeds 679/347/124 consumed in its loop order, "
                         "results_linear_system.py:21-23); single GPU only")
    args = ap.parse_args()
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    import torch
    device = None
    # TMPC_FORCE_PG=1: a one-rank run goes through the process group and the device-side gather as well (the RCCL rehearsal
    # that fits a one-GPU box); the multi-rank launch (torch.distributed.run) always does
    use_pg = world > 1 or os.environ.get("TMPC_FORCE_PG", "0") == "1"
    if use_pg:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    p_loss = np.arange(10) / 10.0                              # :149
    if args.all_controllers:
        controllers = [("tube MPC", False, False), ("extended tube MPC", True, False), ("tracking MPC (non-robust)", False, True)]
    else:
        controllers = [("tracking MPC (non-robust)" if args.rmpc else ("extended tube MPC" if args.extended else "tube MPC"), args.extended, args.rmpc)]
    for label, extended, rmpc in controllers:
        # controller set up in the package: offline sets through the batched LP kernel on this rank's device (0.2 s)
        mpc, model = workloads.make_controller("cartpole", args.N, True, extended=extended, device=local, tracking=rmpc)
        t0 = time.time()
        if args.reference_streams:
            if world != 1:
                raise SystemExit("--reference-streams keeps the reference's draw order and cannot be sharded")
            pl, th, ga, wd = montecarlo.draw_realisations_reference_order(p_loss, args.n_mc, args.T, model["w_bound"])
            out = mpc.run_closed_loop(pl, np.full(args.T, args.ref), th, ga, wd, extended=extended, warm_start=args.warm_start, timing=args.timing)
            table = np.c_[out["tracking_error"], out["tube_violations"], out["not_optimal"]]
            if args.timing:
                table = np.c_[table, out["solve_time_mean"], out["solve_time_max"]]
            pi = np.repeat(np.arange(len(p_loss)), args.n_mc)
        else:
            # one seed for every controller: the same loss patterns and disturbances (per-trajectory streams)
            table, pi = montecarlo.mc_sweep(mpc, model, p_loss, args.n_mc, args.T, args.ref, rank=rank, world=world,
                                            extended=extended, device=device, on_device=not args.host_loop, warm_start=args.warm_start,
                                            timing=args.timing and not args.host_loop, device_rng=args.device_rng, force_collective=use_pg)
        dt = time.time() - t0
        if rank == 0:
            report(label, table, pi, p_loss, dt, world, args)
    if use_pg:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def report(label, table, pi, p_loss, dt, world, args):
    n = len(pi)
    print(f"== {label}: {n} trajectories x {args.T} steps = {n * args.T} solves in {dt:.2f} s on {world} GPU(s): {n * args.T / dt:.3e} MPC steps/s (end to end)")
    print("p_loss  mean tracking error   tube violations   non-optimal solves   infeasible runs")
    for i, p in enumerate(p_loss):
        m = pi == i
        te = table[m, 0]
        dead = int(np.isnan(te).sum())                       # R-MPC: a run whose QP became infeasible stops (NaN tracking error, :297)
        mean_te = float(np.nanmean(te)) if dead < m.sum() else float("nan")
        print(f"{p:5.1f}   {mean_te:.6f}            {int(table[m, 1].sum()):6d}            {int(table[m, 2].sum()):6d}            {dead:6d}")
    if args.timing and table.shape[1] >= 5:                      # results_linear_system.py:305-315, in milliseconds
        mean_ms, max_ms = 1e3 * table[:, 3], 1e3 * table[:, 4]
        print(f"device time per MPC solve (one instance of a batched launch), {n} trajectory means: max of all solves {max_ms.max():.3f} ms, "
              f"95% quantile {np.quantile(mean_ms, 0.95):.3f}, 90% {np.quantile(mean_ms, 0.9):.3f}, 75% {np.quantile(mean_ms, 0.75):.3f}, "
              f"median {np.median(mean_ms):.3f}, mean {mean_ms.mean():.3f} ms")


if __name__ == "__main__":
    main()
