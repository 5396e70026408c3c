#!/usr/bin/env python3
"""Benchmark of the hot path: batched tube-tracking-MPC QP solves on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W      (N > 1, one rank per GPU)

Workload (BASELINE.json configs[1]): linearised cartpole n=4, m=1, horizon N=10, fixed initial state
(results_linear_system.py:120), p_loss = 0.3, batch 4096 (x_k, ref) pairs PER GPU (weak scaling).  The pairs are the
estimates and references the remote controller is actually asked to solve: at start-up every rank builds the controller
(offline sets through the batched LP kernel, 0.2 s) and runs 4096 / 32 seeded closed loops over the lossy network for 32 steps
with its own device solver (workloads.harvest_closed_loop_states): 4096 DISTINCT states, synthetic and seeded.  One step =
one tmpc_solve_batch_device call over the batch, inputs already in HBM (eight seeded random orders of the batch are resident,
the steps go round them).

Metric: QP solves per second (= MPC steps per second), whole job.  The same JSON line carries
  roofline      FP64 flops of the interior-point iterations actually executed (iters per instance come back from the
                device; F_it of SURVEY.md 8(d)) over the average kernel duration from HIP events on the solve stream,
                against the MI355X FP64 peak; kernel name from the library (tmpc_kernel_name);
  cpu_baseline  the CPU oracle (oracle/tmpc_oracle.c, OpenMP over the batch) timed on this box's host cores on a
                bounded sample of the same workload;
  extras        closed loop on the device (cold and warm-started), BASELINE configs[2] (N = 20 extended controller,
                batch 65536, gamma from a real closed loop) and configs[4] (n = 12, m = 4, N = 30, batch 16384; plus a HARD
                leg with an eighth of the states far out), each with its own roofline, and the offline LP stage;
  single_call   the reference's own timing table (results_linear_system.py:305-315: max / 95 / 90 / 75 % / median / mean of
                the wall time of one determine_packet call, in ms) for 1000 calls at batch 1 through tmpc_solve_batch, host
                pointers, copies included -- and for the CPU oracle on one thread;
  closed_loop_small   the reference's own experiment size (results_linear_system.py:64,147-149: N = 20, 10 loss rates x 20
                runs x 250 steps) through the device-resident loop, wall seconds (one fused launch, and a launch pair per step);
  ranks_seen / per_rank_ms   (multi-GPU) an all-reduce of ones and every rank's own timed region and average kernel time.
Order of a one-GPU run: set-up, one solve + the statistics gather (loads torch's kernels), the extras, THEN the W warm-up steps and
the K timed steps.  The first torch kernel of a process costs tens of milliseconds of host time; placed between the warm-up steps and
the timed ones (round 2) it let the card idle, and its clocks need some fifty launches to come back: 4 % on the average kernel
time of K = 20 timed steps (scripts/gpu_ramp.py).  The timed region itself is unchanged: exactly K steps after exactly W, bracketed by
barrier + synchronize, with the statistics gather inside.
Nothing here reads tests/ : the controller set-up lives in the package (workloads.make_controller).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "robust-tracking-mpc-over-lossy-networks_amd"))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6      # MI355X FP64 vector = FP64 matrix peak (AMD datasheet)
HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md


def flops_per_iteration(nv, nc):
    """SURVEY.md 8(d) / BASELINE.md 4: form G'DG (symmetric), Cholesky, two solve pairs,
    four G / G' products."""
    return nc * nv * nv + nv ** 3 / 3.0 + 8.0 * nc * nv + 4.0 * nv * nv


def flops_per_iteration_factored(nv, nd, ncc, kc):
    """The same count with the low-rank block priced at its rank: the ncc rows kept as Hc * Psi cost kc per
    row product and kc^2 per row of G'DG (W = Hc' D Hc), plus Psi' W Psi (2 kc nv^2 ... counted as kc^2 nv + kc nv^2) and the
    four Psi products of an iteration (8 kc nv); the nd general rows and the factorisation as in flops_per_iteration."""
    dense = nd * nv * nv + 8.0 * nd * nv
    fact = ncc * kc * kc + 8.0 * ncc * kc + (kc * kc * nv + kc * nv * nv + 8.0 * kc * nv if ncc else 0.0)
    return dense + fact + nv ** 3 / 3.0 + 4.0 * nv * nv


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


def timing_table(seconds):
    """The statistics the reference prints of its per-call computation times (results_linear_system.py:305-315), in ms."""
    ms = 1e3 * np.asarray(seconds, dtype=np.float64)
    return {"max_ms": float(ms.max()), "q95_ms": float(np.quantile(ms, 0.95)), "q90_ms": float(np.quantile(ms, 0.9)),
            "q75_ms": float(np.quantile(ms, 0.75)), "median_ms": float(np.median(ms)), "mean_ms": float(ms.mean())}


TRAFFIC_PROFILE = os.path.join("profiles", "r04_bench_pmc_summary.txt")


def measured_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the PMC summary committed under profiles/ (rocprofv3 cannot run inside
    this process): FETCH_SIZE + WRITE_SIZE of separate --pmc passes of this very command, in KB as rocprofv3 reports them
    (scripts/profile_round.sh -> scripts/pmc_summary.py writes the file).  None unless the summary is about the kernel
    instantiation the library runs now.  FETCH_SIZE is taken at face value: the x2 of MI355X_MICROARCH.md is calibrated on
    16-byte-per-lane streaming reads, these are 8-byte-per-lane reads of L2-resident model data."""
    try:
        want = kernel_name.replace(" ", "")
        cur, vals = None, {}
        for line in open(os.path.join(ROOT, TRAFFIC_PROFILE)):
            if line.startswith("=="):
                cur = line[2:].strip().replace(" ", "")
            elif cur == want:
                f = line.split()
                if len(f) >= 2 and f[0] in ("FETCH_SIZE", "WRITE_SIZE"):
                    vals[f[0]] = float(f[1]) * 1024.0
        return vals["FETCH_SIZE"] + vals["WRITE_SIZE"] if len(vals) == 2 else None
    except Exception:
        return None


class DeviceBatch:
    """Inputs and outputs of one solve call, resident in HBM."""

    def __init__(self, torch, dev, X, R, gamma, N, nu):
        B, nx = X.shape
        self.B = B
        self.x = torch.from_numpy(np.ascontiguousarray(X)).to(dev)
        self.r = torch.from_numpy(np.ascontiguousarray(R)).to(dev)
        self.g = None if gamma is None else torch.from_numpy(np.ascontiguousarray(gamma.astype(np.uint8))).to(dev)
        self.u = torch.empty((B, N, nu), dtype=torch.float64, device=dev)
        self.x0 = torch.empty((B, nx), dtype=torch.float64, device=dev)
        self.ss = torch.empty((B, nx + nu), dtype=torch.float64, device=dev)
        self.st = torch.empty(B, dtype=torch.int32, device=dev)
        self.it = torch.empty(B, dtype=torch.int32, device=dev)

    def solve(self, native, h):
        native.solve_batch_device(h, self.B, self.x.data_ptr(), self.r.data_ptr(), None if self.g is None else self.g.data_ptr(),
                                  self.u.data_ptr(), self.x0.data_ptr(), self.ss.data_ptr(), None, self.st.data_ptr(), self.it.data_ptr())


def timed_solves(native, torch, h, batch, steps, warmup):
    """-> (wall seconds for `steps` calls, average kernel ms per call from HIP events on the solve stream)."""
    for _ in range(warmup):
        batch.solve(native, h)
    native.synchronize(h)
    torch.cuda.synchronize()
    native.kernel_ms_total(h, reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        batch.solve(native, h)
    native.synchronize(h)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms, n = native.kernel_ms_total(h, reset=True)
    return dt, ms / max(n, 1)


def roofline_entry(native, h, batch, gamma, avg_ms):
    """FP64 roofline of one call: sum over the instances of iters * F_it(nv, nc) of their variant."""
    it = batch.it.cpu().numpy().astype(np.float64)
    st = batch.st.cpu().numpy()
    flops, flops_f, kernels, dims = 0.0, 0.0, [], []
    for v in ([0] if gamma is None else sorted(set(gamma.tolist()))):
        nv, nc, _ = native.get_dims(h, int(v))
        nd, ncc, kc = native.get_factoring(h, int(v))
        on_wave = "solve_kernel" in native.kernel_name(h, int(v))        # (the block kernel keeps every row dense)
        m = np.ones(len(it), bool) if gamma is None else (gamma == v)
        flops += float(it[m].sum()) * flops_per_iteration(nv, nc)
        flops_f += float(it[m].sum()) * (flops_per_iteration_factored(nv, nd, ncc, kc) if on_wave else flops_per_iteration(nv, nc))
        kernels.append(native.kernel_name(h, int(v)))
        dims.append({"variant": int(v), "nv": nv, "nc": nc, "dense_rows": nd, "factored_rows": ncc, "factor_rank": kc,
                     "instances": int(m.sum()), "mean_ipm_iters": float(it[m].mean()) if m.any() else 0.0})
    ach = flops / (avg_ms * 1e-3) / 1e12
    achf = flops_f / (avg_ms * 1e-3) / 1e12
    return ({"bound": "mfma", "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS,
             "kernel": " + ".join(kernels), "avg_kernel_ms": avg_ms, "flops_per_launch": flops,
             "roofline_factored": {"achieved": achf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achf / FP64_PEAK_TFLOPS,
                                   "flops_per_launch": flops_f,
                                   "note": "the low-rank row block priced at its rank kc instead of as dense nv-wide rows "
                                           "(flops_per_iteration_factored); `frac` above is SURVEY.md 8(d)'s dense price"}},
            dims, float((st == 0).mean()), float((st == 2).mean()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4096, help="QP instances per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-closed-loop", action="store_true", help="skip the closed-loop extra (profiling runs: keeps the kernel statistics to the timed steps)")
    ap.add_argument("--no-extras", action="store_true", help="skip the config-3 / config-5 / offline-LP extras")
    ap.add_argument("--only", choices=["config3", "config5"], default=None,
                    help="profiling runs: time ONLY this extra's solve calls (rocprofv3 kernel statistics then belong to it)")
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
    from LinearMPCOverNetworks import _native, montecarlo, workloads

    # TMPC_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks share the devices, the
    # statistics gather goes through host memory); the driver's runs use nccl (= RCCL), one rank per GPU
    backend = os.environ.get("TMPC_BENCH_BACKEND", "nccl")
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if backend == "nccl" and torch.cuda.device_count() < local_world:
        print(f"bench.py: {local_world} ranks on this node but only {torch.cuda.device_count()} visible GPU(s): RCCL needs one GPU per rank "
              "(TMPC_BENCH_BACKEND=gloo rehearses more ranks than GPUs, sharing the devices)", file=sys.stderr)
        sys.exit(3)
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # TMPC_BENCH_FORCE_PG=1: a one-rank run takes the multi-rank code path as well (process group, device-side gather,
    # barrier) -- the RCCL rehearsal that fits a one-GPU box (tests/test_nccl_single_rank.py)
    use_pg = world > 1 or os.environ.get("TMPC_BENCH_FORCE_PG", "0") == "1"
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def config3_extra():
        """BASELINE configs[2]: N = 20, ExtendedTubeTrackingMPC, batch 65536; states and gamma from extended closed loops."""
        mpc3, w3 = workloads.make_controller("cartpole", 20, True, extended=True, device=dev_index)
        X3, R3, G3 = workloads.harvest_closed_loop_states(mpc3, w3, 512, 32, seed=300 + rank, extended=True)
        reps = 65536 // len(X3)
        X3, R3, G3 = np.tile(X3, (reps, 1)), np.tile(R3, (reps, 1)), np.tile(G3, reps)
        p3 = np.random.default_rng(3000 + rank).permutation(len(X3))
        X3, R3, G3 = X3[p3], R3[p3], G3[p3]
        b3 = DeviceBatch(torch, dev, X3, R3, G3, 20, 1)
        dt, ms = timed_solves(_native, torch, mpc3._handle, b3, 20, 2)
        roof, dims, opt, inf = roofline_entry(_native, mpc3._handle, b3, G3, ms)
        return {"value": len(X3) * 20 / dt, "unit": "solves/s", "batch": len(X3), "distinct_states": len(X3) // reps, "steps": 20,
                "ms_per_step": dt / 20 * 1e3, "gamma1_fraction": float(G3.mean()), "optimal_fraction": opt, "infeasible_fraction": inf,
                "variants": dims, "roofline": roof,
                "note": "cartpole N=20, ExtendedTubeTrackingMPC (results_linear_system_with_extendedMPC.py), (x_hat, ref, gamma) from "
                        "512 extended closed loops x 32 steps at p_loss 0.3, tiled to the batch; both problems in one call"}

    def config5_extra(hard=True):
        """BASELINE configs[4]: synthetic n = 12, m = 4, N = 30, batch 16384 (block kernel, MFMA normal matrix).  hard: also time the
        same batch with an eighth of its states far out (profiling runs with --only config5 leave it out, so that the kernel
        statistics and counters of that run belong to the configuration's own workload, as in rounds 2 and 3)."""
        mpc5, w5 = workloads.make_controller("synthetic", 30, True, device=dev_index)
        rng = np.random.default_rng(50 + rank)
        B5 = 16384
        X5 = rng.uniform(-0.5, 0.5, (B5, 12)) * mpc5._Xc.b[:12]                # (x_k, ref) uniform in 0.5 Xc
        R5 = np.zeros((B5, 12))
        R5[:, 0] = rng.uniform(-2, 2, B5)
        b5 = DeviceBatch(torch, dev, X5, R5, None, 30, 4)
        K5 = 10
        dt, ms = timed_solves(_native, torch, mpc5._handle, b5, K5, 1)
        roof, dims, opt, inf = roofline_entry(_native, mpc5._handle, b5, None, ms)
        out5 = {"value": B5 * K5 / dt, "unit": "solves/s", "batch": B5, "steps": K5, "ms_per_step": dt / K5 * 1e3,
                "optimal_fraction": opt, "infeasible_fraction": inf, "variants": dims, "roofline": roof,
                "note": "random stable (A, B), n=12, m=4, N=30, Darup sets, x_k uniform in 0.5 Xc (SURVEY.md 8d)"}
        if not hard:
            return out5
        # the hard end of the same configuration: an eighth of the states scaled by 1.9 (near the boundary of Xc: many active
        # rows, some infeasible) -- the mix of tests/test_full_size.py::test_config5_at_16384
        X5h = X5.copy()
        X5h[: B5 // 8] *= 1.9
        b5h = DeviceBatch(torch, dev, X5h, R5, None, 30, 4)
        dth, msh = timed_solves(_native, torch, mpc5._handle, b5h, K5, 1)
        roofh, dimsh, opth, infh = roofline_entry(_native, mpc5._handle, b5h, None, msh)
        out5["hard"] = {"value": B5 * K5 / dth, "unit": "solves/s", "batch": B5, "steps": K5, "ms_per_step": dth / K5 * 1e3,
                        "optimal_fraction": opth, "infeasible_fraction": infh, "mean_ipm_iters": dimsh[0]["mean_ipm_iters"],
                        "roofline": roofh, "note": "the same batch with X[:B//8] *= 1.9"}
        return out5

    if args.only:
        out = config3_extra() if args.only == "config3" else config5_extra(hard=False)
        if rank == 0:
            print(json.dumps({args.only: out}))
        return

    mpc, w = workloads.make_controller("cartpole", 10, True, device=dev_index)
    h = mpc._handle
    nv, nc, npar = _native.get_dims(h, 0)
    nx, nu, N = 4, 1, 10
    B = args.batch

    # synthetic batch: distinct closed-loop states, every rank its own seeded closed loops
    T_h = 32
    X, R, _ = workloads.harvest_closed_loop_states(mpc, w, (B + T_h - 1) // T_h, T_h, seed=1000 + rank)
    X, R = X[:B], R[:B]
    distinct = len(np.unique(np.c_[X, R], axis=0))
    # The harvest is time-step-major (all transients first), and with two instances per resident wave the launch time depends
    # on which instances happen to come last (0.53 ... 0.76 ms over random orders of these very states).  Eight seeded random
    # orders of the same 4096 states are kept in HBM and the steps go round them, so `value` is the mean over orders.
    NORD = 8
    batches = []
    for k in range(NORD):
        perm = np.random.default_rng(2000 + 16 * rank + k).permutation(B)
        batches.append(DeviceBatch(torch, dev, X[perm], R[perm], None, N, nu))
    batch = batches[0]
    torch.cuda.synchronize()

    def fence():
        _native.synchronize(h)
        torch.cuda.synchronize()
        if use_pg:
            dist.barrier()
            torch.cuda.synchronize()

    def gather_stats():
        # the only exchange of the path: per-trajectory statistics, gathered once per sweep
        # (RCCL all-gather over xGMI; 8 B per trajectory, latency-bound)
        stats = torch.stack([batch.st, batch.it], dim=1).contiguous()
        if backend != "nccl":
            stats = stats.cpu()
        return montecarlo.gather_statistics(stats, world * B, rank, world, force_collective=use_pg)

    def extras():
        """The other sections of the line (closed loop, pipelined handles, configs 3 and 5, offline LPs), one GPU only.  They run BEFORE
        the headline steps: the card then goes into the W warm-up steps and the K timed ones from seconds of the same kind of
        load instead of from the idle gap of the set-up (kernel times of the first steps after an idle half second are 2-4 % up)."""
        ex = {}
        if world == 1 and not args.no_closed_loop:
            # the same kernel inside the device-resident closed loop over the lossy network (tmpc_mc_run): every step is
            # the solve + the estimator / actuator / plant state machines, 4096 trajectories, p_loss = 0.3 (configs[1]);
            # reference step at t = 0 and again half way, so that transients and settled phases are both in the run
            Tcl = 100
            th, ga, wd = montecarlo.draw_realisations(B, Tcl, w["w_bound"], seed=99)
            pl = np.full(B, 0.3)
            ref_cl = np.where(np.arange(Tcl) < Tcl // 2, 0.5, -0.5)
            mpc.run_closed_loop(pl[:64], ref_cl, th[:64], ga[:64], wd[:64])          # warm-up
            cl_out = {}
            for warm in (False, True):
                leg = {}
                for mode in ("off", None):        # a launch pair per step, then the library's own choice (tmpc_mc_set_fused: AUTO)
                    tcl = time.perf_counter()
                    cl = mpc.run_closed_loop(pl, ref_cl, th, ga, wd, warm_start=warm, fused=mode)
                    tcl = time.perf_counter() - tcl
                    if mode == "off":
                        leg["per_step_launches"] = {"value": B * Tcl / tcl, "launches": 2 * Tcl}
                        ref_run = cl
                        continue
                    leg.update({"value": B * Tcl / tcl, "unit": "MPC steps/s", "fused": bool(cl["fused"]), "launches": 1 if cl["fused"] else 2 * Tcl,
                                "same_as_per_step_bit_for_bit": bool(all(np.array_equal(cl[k], ref_run[k], equal_nan=True) for k in
                                                                         ("err2", "x_final", "tube_violations", "not_optimal", "iters_sum"))),
                                "tube_violations": int(cl["tube_violations"].sum()),
                                "non_optimal_solves": int(cl["not_optimal"].sum()), "mean_ipm_iters": float(cl["iters_mean"]),
                                "tracking_error_mean": float(cl["tracking_error"].mean())})
                cl_out["warm" if warm else "cold"] = leg
            ex["closed_loop"] = {"trajectories": B, "steps": Tcl, "p_loss": 0.3, **cl_out,
                                  "note": "end to end incl. upload of the realisations and download of the statistics; `value`: tmpc_mc_run as the "
                                          "library runs it by default -- ONE launch for the sweep, a wavefront keeps its trajectory for all steps "
                                          "(closed_loop_kernel) --, per_step_launches: one solve launch + one state-machine launch per step; warm = "
                                          "every solve first tries the working set of the trajectory's previous step in the exact refinement"}
        if world == 1 and not args.no_extras:
            # The reference's own performance self-description (results_linear_system.py:305-315): wall time of ONE
            # determine_packet call, max / quantiles / median / mean in ms.  Here: 1000 calls at batch 1 through tmpc_solve_batch
            # (host pointers in and out, copies and the synchronisation included), states drawn from the bench batch.
            ii = np.random.default_rng(11).integers(0, B, 1000)
            for k in range(20):
                mpc.determine_packet(X[ii[k]], R[ii[k]], 0)
            mpc.reset_computational_times()
            for k in ii:
                mpc.determine_packet(X[k], R[k], 0)
            ex["single_call"] = {"device": timing_table(mpc.get_computational_times()), "calls": len(ii), "batch": 1,
                                 "note": "TubeTrackingMPC.determine_packet (TubeTrackingMPC.py:196-209) -> tmpc_solve_batch, one QP per "
                                         "call, host buffers, copies included; statistics of results_linear_system.py:305-315 "
                                         "(the reference's histogram spans 2.5-20 ms)"}
            mpc.reset_computational_times()
            # the reference's experiment at its own size (results_linear_system.py:64, 147-149: N = 20, 10 loss rates x 20 runs x
            # 250 steps), device-resident loop, realisations drawn on the host as the reference does: 200 trajectories are
            # 200 of the card's 2048 resident waves, so this is a launch-latency figure (two launches per step), not throughput
            mpc20, w20 = workloads.make_controller("cartpole", 20, True, device=dev_index)
            pl20 = np.repeat(np.arange(10) / 10.0, 20)
            th20, ga20, wd20 = montecarlo.draw_realisations(len(pl20), 250, w20["w_bound"], seed=7)
            ref20 = 0.5 * np.ones(250)
            mpc20.run_closed_loop(pl20[:8], ref20[:20], th20[:8, :20], ga20[:8, :20], wd20[:8, :20])          # warm-up
            small = {}
            for warm in (False, True):
                leg = {}
                for mode in ("off", None):
                    ts = time.perf_counter()
                    cl = mpc20.run_closed_loop(pl20, ref20, th20, ga20, wd20, warm_start=warm, fused=mode)
                    ts = time.perf_counter() - ts
                    if mode == "off":
                        leg["per_step_launches"] = {"wall_s": ts, "ms_per_step": ts / 250 * 1e3, "launches": 500}
                        continue
                    leg.update({"wall_s": ts, "ms_per_step": ts / 250 * 1e3, "MPC_steps_per_s": len(pl20) * 250 / ts, "fused": bool(cl["fused"]),
                                "launches": 1 if cl["fused"] else 500, "tube_violations": int(cl["tube_violations"].sum()),
                                "non_optimal_solves": int(cl["not_optimal"].sum())})
                small["warm" if warm else "cold"] = leg
            ex["closed_loop_small"] = {"trajectories": len(pl20), "steps": 250, "N": 20, **small,
                                       "note": "10 loss rates x 20 runs x 250 steps, N = 20 (results_linear_system.py:64,147-149), "
                                               "tmpc_mc_run: one launch for the whole experiment (200 wavefronts, each with its trajectory for 250 "
                                               "steps: the figure is 250 x the latency of one solve + state machines); per_step_launches: a solve "
                                               "launch + a state-machine launch per step"}
            del mpc20
            # ... and the reference's second experiment at its own size (results_linear_system_with_extendedMPC.py:133-147: the extended
            # controller, N = 20, 10 x 20 x 250): two problems, chosen per step by the arrival flag -- one launch per problem and step
            # with the state machines inside (tmpc_mc_last_fused = 2), against two solve launches + a state-machine launch
            mpc20x, _ = workloads.make_controller("cartpole", 20, True, extended=True, device=dev_index)
            mpc20x.run_closed_loop(pl20[:8], ref20[:20], th20[:8, :20], ga20[:8, :20], wd20[:8, :20], extended=True)          # warm-up
            smallx = {}
            for warm in (False, True):
                leg = {}
                for mode in ("off", None):
                    ts = time.perf_counter()
                    cl = mpc20x.run_closed_loop(pl20, ref20, th20, ga20, wd20, extended=True, warm_start=warm, fused=mode)
                    ts = time.perf_counter() - ts
                    if mode == "off":
                        leg["three_launches_per_step"] = {"wall_s": ts, "ms_per_step": ts / 250 * 1e3}
                        continue
                    leg.update({"wall_s": ts, "ms_per_step": ts / 250 * 1e3, "MPC_steps_per_s": len(pl20) * 250 / ts, "loop_mode": int(cl["loop_mode"]),
                                "launches": 500 if cl["loop_mode"] == 2 else 750, "tube_violations": int(cl["tube_violations"].sum()),
                                "non_optimal_solves": int(cl["not_optimal"].sum())})
                smallx["warm" if warm else "cold"] = leg
            ex["closed_loop_small"]["extended"] = {**smallx, "note": "ExtendedTubeTrackingMPC + RobustEstimator + ConsistentActuator, same size"}
            del mpc20x
            # offline stage extra: support-function LPs over this workload's terminal set in one launch (tmpc_lp_batch)
            Xf = mpc._Xf
            dirs = np.random.default_rng(7).standard_normal((65536, Xf.A.shape[1]))
            _native.lp_batch(Xf.A, Xf.b, dirs[:256])
            tlp = time.perf_counter()
            lp = _native.lp_batch(Xf.A, Xf.b, dirs)
            tlp = time.perf_counter() - tlp
            ex["offline_lp"] = {"value": len(dirs) / tlp, "unit": "LP/s", "rows": int(Xf.A.shape[0]), "dim": int(Xf.A.shape[1]),
                                 "batch": len(dirs), "solved": int((lp["status"] == 0).sum()),
                                 "note": "support LPs over the terminal set, host buffers in and out (set-up stage, DESIGN.md 7a)"}
            ex["config3"] = config3_extra()
            ex["config5"] = config5_extra()
        if not args.no_extras:
            # (on every rank, and last of the extras: the headline's warm-up steps then follow a dense sequence of launches of its
            # own kernel for every N, so that the per-N values the scaling efficiency is computed from are measured alike)
            # two handles (two streams) taking turns over the same batches: a launch of 4096 ends with its slowest instance
            # (two instances per resident wave), and the tail of one launch overlaps with the head of the next when it is
            # on another stream -- the throughput a server sees that pipelines its batches.  `value` above stays the
            # one-stream figure, whose kernel durations the roofline entry prices.
            mpc_b, _ = workloads.make_controller("cartpole", 10, True, device=dev_index)
            twin = [DeviceBatch(torch, dev, X[p_], R[p_], None, N, nu) for p_ in
                    (np.random.default_rng(2100 + k).permutation(B) for k in range(NORD))]
            hs, bs = (mpc._handle, mpc_b._handle), (batches, twin)
            for k in range(2 * NORD):
                bs[k % 2][k % NORD].solve(_native, hs[k % 2])
            for h_ in hs:
                _native.synchronize(h_)
            tp = time.perf_counter()
            for k in range(args.steps):
                bs[k % 2][k % NORD].solve(_native, hs[k % 2])
            for h_ in hs:
                _native.synchronize(h_)
            tp = time.perf_counter() - tp
            if use_pg:
                tpm = torch.tensor([tp], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
                dist.all_reduce(tpm, op=dist.ReduceOp.MAX)
                tp = float(tpm.item())
            ex["pipelined"] = {"value": world * B * args.steps / tp, "unit": "solves/s", "handles": 2, "steps": args.steps,
                                "ms_per_step": tp / args.steps * 1e3,
                                "note": "same batches, two handles per GPU on their own streams taking turns (tails of successive launches "
                                        "overlap); whole job, slowest rank"}
        return ex

    # torch's own kernels and the collective are loaded / set up here, NOT between the warm-up steps and the timed ones: the first
    # torch.stack of a process takes tens of milliseconds of host time, the card idles, and its clocks need some fifty launches
    # (25 ms) to come back -- 4 % on the average kernel time of K = 20 timed steps (scripts/gpu_ramp.py)
    batch.solve(_native, h)
    _native.synchronize(h)
    gather_stats()
    fence()
    extras_out = extras()
    torch.cuda.synchronize()

    for i in range(args.warmup):
        batches[i % NORD].solve(_native, h)
    fence()
    _native.kernel_ms_total(h, reset=True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        batches[i % NORD].solve(_native, h)
    _native.synchronize(h)
    stats_all = gather_stats()
    fence()
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed
    if use_pg:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    kern_ms, launches = _native.kernel_ms_total(h, reset=True)
    # Evidence that the collective really joined `world` ranks, each on a GPU of its own: an all-reduce of ones, and every
    # rank's own timed region and average kernel duration (outside the timed region)
    ranks_seen, per_rank = 1, [[elapsed_local * 1e3 / args.steps, kern_ms / max(launches, 1), float(dev_index)]]
    if use_pg:
        cdev = dev if backend == "nccl" else "cpu"
        ones = torch.ones(1, dtype=torch.float64, device=cdev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(round(float(ones.item())))
        mine = torch.tensor(per_rank[0], dtype=torch.float64, device=cdev)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [a.cpu().tolist() for a in allr]
        if ranks_seen != world:
            print(f"bench.py: the all-reduce saw {ranks_seen} ranks, expected {world}", file=sys.stderr)
            sys.exit(4)
    # spread of the launch time over the eight batch orders (outside the timed region: one launch each, HIP events)
    _native.synchronize(h)
    _native.kernel_ms_total(h, reset=True)
    order_ms = []
    for k in range(NORD):
        batches[k].solve(_native, h)
        _native.synchronize(h)
        ms_k, n_k = _native.kernel_ms_total(h, reset=True)
        order_ms.append(ms_k / max(n_k, 1))
    stats_np = stats_all.cpu().numpy()
    status_all, iters_all = stats_np[:, 0], stats_np[:, 1]

    if rank == 0:
        value = world * B * args.steps / elapsed
        avg_ms = kern_ms / max(launches, 1)
        roof, dims, opt, inf = roofline_entry(_native, h, batch, None, avg_ms)
        bytes_solve = 8 * (2 * nx) + 8 * (N * nu + nx + nx + nu) + 8      # inputs + the outputs this call writes + status/iters
        roof["traffic"] = measured_traffic(roof["kernel"])
        roof["traffic_source"] = TRAFFIC_PROFILE
        roof["launch_ms_by_order"] = {"min": min(order_ms), "max": max(order_ms), "mean": float(np.mean(order_ms)),
                                       "note": "one launch per batch order, same 4096 states (two instances per resident wave: the "
                                               "launch ends with its slowest wave)"}
        roof["note"] = ("bound is FP64 arithmetic (vector ALU + the FP64 MFMA normal-matrix pass; 78.6 TFLOP/s is the peak of "
                        "either), not HBM: algorithmic HBM bytes are %d B/solve" % bytes_solve)
        roof["hbm"] = {"achieved": bytes_solve * B / (avg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                       "frac": bytes_solve * B / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        out = {
            "metric": "QP solves/sec (= MPC steps/sec) at batch",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "cartpole n=4 m=1 N=10 tube-tracking QP, fixed x0, p_loss 0.3, batch 4096 per GPU: the (x_hat, ref) "
                                   "pairs of 128 seeded closed loops x 32 steps over the lossy network (BASELINE configs[1])",
                       "batch_per_gpu": B, "distinct_states": int(distinct), "nv": nv, "nc": nc,
                       "parallelism": f"trajectory-sharded x{world}",
                       "mean_ipm_iters": float(iters_all.mean()), "optimal_fraction": float((status_all == 0).mean()),
                       "trivial_fraction": float((iters_all == 0).mean())},
            "roofline": roof,
            "ranks_seen": ranks_seen,
            "per_rank_ms": {"ms_per_step": [r_[0] for r_ in per_rank], "avg_kernel_ms": [r_[1] for r_ in per_rank],
                            "device_index": [int(r_[2]) for r_ in per_rank], "backend": backend if use_pg else "none",
                            "note": "every rank's own timed region / K and average solve-kernel duration (HIP events); "
                                    "ms_per_step above is the slowest rank's"},
            # what the card ran between the set-up and the W warm-up steps (the headline depends on it by 2-4 %: clock ramp,
            # scripts/gpu_ramp.py): "extras" = seconds of the other sections' launches, "none" = the set-up's idle time
            "preload": ("none" if args.no_extras and (args.no_closed_loop or world > 1) else
                        "extras" if not args.no_extras else "closed_loop"),
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle.oracle import Oracle
            orc = Oracle(mpc._problem_dict())
            cores = host_cores()
            ns = 131072
            ii = np.random.default_rng(5).integers(0, B, ns)
            Xc, Rc = X[ii].copy(), R[ii].copy()
            orc.solve(Xc[:256], Rc[:256], nthreads=cores)
            tc = time.perf_counter()
            oc = orc.solve(Xc, Rc, nthreads=cores)
            tc = time.perf_counter() - tc
            out["cpu_baseline"] = {"value": ns / tc, "unit": "solves/s", "cores": cores, "kind": "port",
                                   "sample": f"{ns} instances drawn from the same 4096 states, oracle/tmpc_oracle.c (IPM + refinement), "
                                             f"OpenMP over the batch, {tc:.2f} s wall, mean iters {oc['iters'].mean():.2f}"}
            # the same solver on ONE host thread (what the reference's loop has: one solve at a time)
            n1 = 8192
            t1 = time.perf_counter()
            o1 = orc.solve(Xc[:n1], Rc[:n1], nthreads=1)
            t1 = time.perf_counter() - t1
            out["cpu_baseline"]["single_thread"] = {"value": n1 / t1, "unit": "solves/s", "cores": 1, "kind": "port",
                                                    "sample": f"the first {n1} of those instances, one thread, {t1:.2f} s wall, "
                                                              f"mean iters {o1['iters'].mean():.2f}"}
            # the reference's timing table for the CPU solver: 1000 single-QP calls on one thread (the oracle through ctypes)
            tsc = []
            for k in np.random.default_rng(11).integers(0, B, 1000):
                a0 = time.perf_counter()
                orc.solve(X[k:k + 1], R[k:k + 1], nthreads=1)
                tsc.append(time.perf_counter() - a0)
            out["cpu_baseline"]["single_call"] = dict(timing_table(tsc), calls=len(tsc), kind="port",
                                                      note="oracle/tmpc_oracle.c, one QP per call, one thread")
            if "single_call" in extras_out:
                extras_out["single_call"]["cpu_oracle_one_thread"] = out["cpu_baseline"]["single_call"]
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
        out.update(extras_out)
        print(json.dumps(out))
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
