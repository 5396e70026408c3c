"""Sharding of a Monte-Carlo sweep over ranks (one process per GPU) and the one collective of
the path: gathering per-trajectory statistics.

The reference runs `for i in p_loss: for l_mc in N_MC: for t in T:` in one Python process
(results_linear_system.py:165-209); trajectories never interact (estimator/actuator are
re-created per run, :186-188), so they are sharded with no data-path collective.  What is
exchanged is what the script aggregates afterwards (:291 tracking error, :268-270 failure
counts, :305-315 timing/iteration statistics): a few numbers per trajectory, once per sweep.
"""
from __future__ import annotations

import numpy as np


def trajectory_table(p_loss, n_mc: int):
    """Global trajectory index -> (p_loss index, seed index), p_loss-minor so that every
    contiguous shard sees all loss rates (balanced iteration counts)."""
    p_loss = np.asarray(p_loss, dtype=np.float64)
    n = len(p_loss) * int(n_mc)
    g = np.arange(n)
    return g % len(p_loss), g // len(p_loss)


def shard_bounds(n_items: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of rank; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_statistics(local, n_total: int, rank: int, world: int, group=None, force_collective: bool = False):
    """All-gather of a (n_local, k) tensor of per-trajectory statistics into the global
    (n_total, k) table, identical on every rank.  Backend-agnostic: `nccl` (= RCCL over xGMI) on
    the GPUs, `gloo` in the CPU tests.  Shards of unequal size are padded to the largest.
    force_collective: a one-rank run goes through the collective as well (needs an initialised process group;
    the single-GPU rehearsal of the RCCL path)."""
    import torch
    import torch.distributed as dist
    if world == 1 and not force_collective:
        return local
    sizes = [shard_bounds(n_total, r, world)[1] - shard_bounds(n_total, r, world)[0] for r in range(world)]
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank} holds {local.shape[0]} rows, expected {sizes[rank]}")
    m = max(sizes)
    pad = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[:s] for o, s in zip(out, sizes)], dim=0)


# --------------------------------------------------------------------------- closed loop
def draw_realisations(n_traj: int, T: int, w_bound, seed: int = 20240301, first: int = 0):
    """Per-trajectory random streams (uniforms for theta, gamma and the disturbance), so that a
    trajectory's realisation does not depend on how the sweep is sharded.  Trajectory g uses
    SeedSequence(seed, spawn_key=(g,)).  (The reference draws from three shared generators in loop
    order, results_linear_system.py:21-23,218-233; that order cannot be kept under sharding.)"""
    w_bound = np.asarray(w_bound, dtype=np.float64)
    th = np.empty((n_traj, T))
    ga = np.empty((n_traj, T))
    w = np.empty((n_traj, T, w_bound.size))
    for i in range(n_traj):
        rng = np.random.default_rng(np.random.SeedSequence(seed, spawn_key=(first + i,)))
        th[i] = rng.uniform(size=T)
        ga[i] = rng.uniform(size=T)
        w[i] = rng.uniform(-1.0, 1.0, size=(T, w_bound.size)) * w_bound
    return th, ga, w


# Philox4x64-10 (Salmon et al., SC'11), the numpy twin of the device generator in csrc/tmpc_mc.hip (tmpc_mc_set_device_rng).
_PHILOX_M0, _PHILOX_M1 = 0xD2E7470EE14C6C93, 0xCA5A826395121157
_PHILOX_W0, _PHILOX_W1 = 0x9E3779B97F4A7C15, 0xBB67AE8584CAA73B


def _mulhilo64(a, b: int):
    """(high, low) 64-bit halves of a * b for a uint64 array a and a 64-bit constant b."""
    m32 = np.uint64(0xFFFFFFFF)
    s32 = np.uint64(32)
    a0, a1 = a & m32, a >> s32
    b0, b1 = np.uint64(b & 0xFFFFFFFF), np.uint64(b >> 32)
    p00, p01, p10, p11 = a0 * b0, a0 * b1, a1 * b0, a1 * b1
    carry = ((p00 >> s32) + (p01 & m32) + (p10 & m32)) >> s32
    return p11 + (p01 >> s32) + (p10 >> s32) + carry, a * np.uint64(b)


def philox4x64(c0, c1, k0, k1):
    """Ten rounds of Philox-4x64 on the counter (c0, c1, 0, 0) with the key (k0, k1), vectorised: arrays that broadcast
    against each other -> (4, ...) uint64.  numpy.random.Philox(key=[k0, k1], counter=[c0 - 1, c1, 0, 0]).random_raw(4) gives
    the same four words (numpy increments the counter before it generates; tests/test_condense.py pins this)."""
    with np.errstate(over="ignore"):
        c0, c1, k0, k1 = np.broadcast_arrays(*(np.asarray(v, dtype=np.uint64) for v in (c0, c1, k0, k1)))
        c = [c0.copy(), c1.copy(), np.zeros_like(c0), np.zeros_like(c0)]
        k0, k1 = k0.copy(), k1.copy()
        for _ in range(10):
            hi0, lo0 = _mulhilo64(c[0], _PHILOX_M0)
            hi1, lo1 = _mulhilo64(c[2], _PHILOX_M1)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            k0 = k0 + np.uint64(_PHILOX_W0)
            k1 = k1 + np.uint64(_PHILOX_W1)
    return np.stack(c)


def draw_realisations_philox(n_traj: int, T: int, w_bound, seed: int = 20240301, first: int = 0):
    """The realisations tmpc_mc_run draws on the device with tmpc_mc_set_device_rng(seed, first, w_bound), reproduced on the
    host (include/tmpc.h): trajectory g = first + i, step t: Philox4x64-10 with key (seed, g), counter (t, j, 0, 0);
    block 0 = [theta, gamma, w_0, w_1], block j = w_{4j-2} .. w_{4j+1}; u = (x >> 11) 2^-53; w_i = w_bound_i (2 u - 1).
    Like draw_realisations, a trajectory's stream does not depend on how the sweep is sharded."""
    w_bound = np.asarray(w_bound, dtype=np.float64).reshape(-1)
    nx = w_bound.size
    g = (np.uint64(first) + np.arange(n_traj, dtype=np.uint64))[:, None]
    t = np.arange(T, dtype=np.uint64)[None, :]
    nblk = (nx + 2 + 3) // 4
    words = np.concatenate([philox4x64(t, np.uint64(j), np.uint64(seed), g) for j in range(nblk)], axis=0)     # (4 nblk, n, T)
    u = (words >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    th, ga = u[0], u[1]
    w = np.moveaxis(u[2:2 + nx], 0, -1) * 2.0 - 1.0
    return np.ascontiguousarray(th), np.ascontiguousarray(ga), np.ascontiguousarray(w * w_bound)


def draw_realisations_reference_order(p_loss, n_mc: int, T: int, w_bound, seeds=(679, 347, 124)):
    """The realisations of the reference's own experiment: three shared generators -- disturbance 679, gamma 347,
    theta 124 (results_linear_system.py:21-23) -- consumed in loop order over (loss rate i, run l_mc, step t): one theta
    and one gamma uniform per step t >= 1 (:218-226, the first transmission always succeeds, :211-214) and nx disturbance
    components per step (:229-233).  Only valid for the whole sweep on one rank (the order is the point).

    Returns (p (B,), th (B,T), ga (B,T), w (B,T,nx)), B = len(p_loss) * n_mc in (i, l_mc) order; the t = 0 entries
    of th/ga are 1.0 (never below a loss probability)."""
    p_loss = np.asarray(p_loss, dtype=np.float64)
    w_bound = np.asarray(w_bound, dtype=np.float64)
    rng_w, rng_gamma, rng_theta = (np.random.default_rng(sd) for sd in seeds)
    nb = len(p_loss) * int(n_mc)
    th = np.ones((nb, T))
    ga = np.ones((nb, T))
    th[:, 1:] = rng_theta.uniform(size=(nb, T - 1))
    ga[:, 1:] = rng_gamma.uniform(size=(nb, T - 1))
    w = rng_w.uniform(-w_bound, w_bound, size=(nb, T, w_bound.size))
    return np.repeat(p_loss, int(n_mc)), th, ga, w


def run_remote_tube_mpc(packets_fn, A, B, K, K_plant, N, Z, p_loss, ref, th_u, ga_u, w, x0=None, extended: bool = False,
                        plant=None, capture=None, observer=None):
    """Closed loop of the remote tube-based MPC over a lossy network for a batch of trajectories:
    the body of the reference's Monte-Carlo loop (results_linear_system.py:209-259, 291) with the
    per-trajectory objects replaced by the batched state machines and the QP solves of one time
    step done by ONE call of `packets_fn(x_hat (B,nx), ref_t (B,nx)[, gamma (B,)]) -> (U_t (B,nu,N+1),
    x_nom0, status (B,))` -- normally `TubeTrackingMPC.determine_packets`, i.e. one kernel launch.

    extended=True is the loop of results_linear_system_with_extendedMPC.py:247-378: the controller is an
    ExtendedTubeTrackingMPC that is told whether the previous plant packet arrived (gamma_{t-1}, :276), the
    estimator is the RobustEstimator (it also stores x_nom_0, :279) and the actuator adopts x_nom_0 (:133-147).

    p_loss (B,), ref (T,) or (B,T) position reference, th_u/ga_u (B,T) uniforms, w (B,T,nx) disturbances.
    plant: None = the linear model x+ = A x + B u + w (:248); or a callable (x (B,nx), u (B,nu)) -> x+ (w is added to it),
    e.g. workloads.cartpole_step for the nonlinear cart-pole of results_nonlinear_system.py.
    capture: index of one trajectory whose x_t, nominal state of the tube check and u_t are recorded (the scripts' sample run,
    :298-301) -> 'x_traj' (T, nx), 'x_nom_traj' (T, nx), 'u_traj' (T, nu).
    observer: optional callable (t, {'s', 'Theta', 'u'}) called after the actuator of step t (copies of its s_t, Theta_t and u_t).
    Returns a dict of per-trajectory statistics."""
    from .Estimator import BatchedEstimator
    from .SmartActuator import BatchedConsistentActuator
    A = np.asarray(A, dtype=np.float64)
    Bm = np.asarray(B, dtype=np.float64)
    nb, T = th_u.shape
    nx = A.shape[0]
    p_loss = np.asarray(p_loss, dtype=np.float64).reshape(nb)
    ref = np.asarray(ref, dtype=np.float64)          # (T,) shared by the batch, or (B, T) per trajectory

    def ref_at(t):
        return ref[t] if ref.ndim == 1 else ref[:, t]
    x = np.zeros((nb, nx)) if x0 is None else np.array(x0, dtype=np.float64).reshape(nb, nx)
    cap = None if capture is None else dict(x_traj=np.zeros((T, nx)), x_nom_traj=np.zeros((T, nx)), u_traj=np.zeros((T, Bm.shape[1])))
    est = BatchedEstimator(A, Bm, K, x, N, K_plant=K_plant if extended else None, robust=extended)
    act = BatchedConsistentActuator(A, Bm, K, K_plant, x, is_extended_MPC_used=extended)
    err2 = np.zeros(nb)
    err2_phys, n_phys = np.zeros(nb), 0
    tube_viol = np.zeros(nb, dtype=np.int32)
    not_optimal = np.zeros(nb, dtype=np.int32)
    consistent_err = 0.0
    U_prev = x0_prev = None
    gamma = np.ones(nb, dtype=np.int64)
    for t in range(T):
        theta = np.where(th_u[:, t] < p_loss, 0, 1) if t > 0 else np.ones(nb, dtype=np.int64)     # :211-226, strict <
        r_t = np.zeros((nb, nx))
        r_t[:, 0] = ref_at(t)
        q_t = est.get_qt()
        if extended:
            U_t, x_nom_0, status = packets_fn(est.get_estimate(), r_t, gamma.astype(np.uint8))     # RLX:276, gamma of step t-1
        else:
            U_t, x_nom_0, status = packets_fn(est.get_estimate(), r_t)                             # :240
        bad = status >= 2
        not_optimal += (status != 0)
        if bad.any():
            # the reference's tube branch has no handling for a failed solve (it would raise); here the
            # packet of such a trajectory is treated as lost and its previous sequence stays in use
            U_t = np.where(bad[:, None, None], U_prev if U_prev is not None else 0.0, U_t)
            x_nom_0 = np.where(bad[:, None], x0_prev if x0_prev is not None else 0.0, x_nom_0)
            theta = np.where(bad, 0, theta)
        U_prev, x0_prev = U_t, x_nom_0
        est.store(U_t)                                                                             # :242
        if extended:
            est.store_x_nom_0(x_nom_0)                                                             # RLX:279
        x_nom_now = act.x_nom.copy()       # the nominal state the scripts test against: column t of x_nom_traj, i.e. BEFORE process_packet
        u, pkt = act.process(U_t, q_t, x, theta, x_nom_0 if extended else None)                    # :244
        if observer is not None:
            observer(t, dict(s=np.array(act.s).copy(), Theta=np.array(act.Theta).copy(), u=np.array(u).copy()))
        err2 += (x[:, 0] - ref_at(t)) ** 2 + np.sum(x[:, 1:] ** 2, axis=1)                            # :291 (x_t, t = 0..T-1)
        # :258 / results_linear_system_with_extendedMPC.py:310-318,331-333 -- x_traj[:, t] - x_nom_traj[:, t]: the nominal state
        # appended after the PREVIOUS step's process_packet, so for the extended controller the state before this step's
        # adoption of x_nom_0 (SmartActuator.py:219-222); for the plain tube MPC the two coincide
        tube_viol += ~np.asarray(Z.contains((x - x_nom_now).T)).reshape(nb)
        if cap is not None:
            cap["x_traj"][t], cap["x_nom_traj"][t], cap["u_traj"][t] = x[capture], x_nom_now[capture], u[capture]
        if plant is not None and hasattr(plant, "trace"):
            xs = plant.trace(x, u)                                    # results_nonlinear_system.py:332-361: error over x_traj[:, 0:-1] at 500 Hz
            err2_phys += np.sum((xs[:-1, :, 0] - ref_at(t)) ** 2 + np.sum(xs[:-1, :, 1:] ** 2, axis=2), axis=0)
            n_phys += xs.shape[0] - 1
            x = xs[-1] + w[:, t]
        else:
            x = (x @ A.T + u @ Bm.T if plant is None else plant(x, u)) + w[:, t]                  # :248
        gamma = np.where(ga_u[:, t] < p_loss, 0, 1) if t > 0 else np.ones(nb, dtype=np.int64)     # :218-226
        est.update(pkt, gamma)                                                                     # :254
        # Proposition 1 of the paper: whenever the actuator is consistent and the plant packet arrives,
        # the estimate equals the nominal plant state
        ok = (act.Theta == 1) & (gamma == 1)
        if ok.any():
            consistent_err = max(consistent_err, float(np.max(np.abs(est.x_hat[ok] - act.x_nom[ok]))))
    out = dict(tracking_error=np.sqrt(err2) / T, tube_violations=tube_viol, not_optimal=not_optimal,
               consistent_estimate_error=consistent_err, x_final=x)
    if n_phys:
        out["tracking_error_physics"] = np.sqrt(err2_phys) / n_phys
    if cap is not None:
        out.update(cap)
    return out


def run_remote_tracking_mpc(packets_fn, A, B, K, N, p_loss, ref, th_u, ga_u, w, x0=None):
    """Closed loop of the non-robust comparator (R-MPC) over the lossy network: TrackingMPC + Estimator + plain
    SmartActuator (results_linear_system.py:198-205, 262-287).  A trajectory whose solve is infeasible stops there
    (track_feasible = False, :268-270) and reports a NaN tracking error (:297).  Same conventions as
    run_remote_tube_mpc otherwise."""
    from .Estimator import BatchedEstimator
    from .SmartActuator import BatchedConsistentActuator
    A = np.asarray(A, dtype=np.float64)
    Bm = np.asarray(B, dtype=np.float64)
    nb, T = th_u.shape
    nx = A.shape[0]
    p_loss = np.asarray(p_loss, dtype=np.float64).reshape(nb)
    ref = np.asarray(ref, dtype=np.float64)          # (T,) shared by the batch, or (B, T) per trajectory

    def ref_at(t):
        return ref[t] if ref.ndim == 1 else ref[:, t]
    x = np.zeros((nb, nx)) if x0 is None else np.array(x0, dtype=np.float64).reshape(nb, nx)
    est = BatchedEstimator(A, Bm, K, x, N)
    act = BatchedConsistentActuator(A, Bm, K, np.zeros_like(np.atleast_2d(K)), x)     # no nominal model: x_nom := x each step
    err2 = np.zeros(nb)
    dead = np.zeros(nb, dtype=bool)
    not_optimal = np.zeros(nb, dtype=np.int32)
    U_prev = None
    for t in range(T):
        theta = np.where(th_u[:, t] < p_loss, 0, 1) if t > 0 else np.ones(nb, dtype=np.int64)
        r_t = np.zeros((nb, nx))
        r_t[:, 0] = ref_at(t)
        q_t = est.get_qt()
        U_t, _, status = packets_fn(est.get_estimate(), r_t)
        newly = ~dead & (status >= 2)
        not_optimal += (~dead & (status != 0))
        dead |= newly
        U_t = np.where(np.isfinite(U_t), U_t, 0.0 if U_prev is None else U_prev)       # keeps the frozen trajectories' state machines NaN-free
        U_prev = U_t
        est.store(U_t)
        act.x_nom = x.copy()
        u, pkt = act.process(U_t, q_t, x, theta)
        pkt = {"x_t": x.copy(), "s_t": pkt["s_t"]}
        err2 += np.where(dead, 0.0, (x[:, 0] - ref_at(t)) ** 2 + np.sum(x[:, 1:] ** 2, axis=1))
        x = np.where(dead[:, None], x, x @ A.T + u @ Bm.T + w[:, t])
        gamma = np.where(ga_u[:, t] < p_loss, 0, 1) if t > 0 else np.ones(nb, dtype=np.int64)
        est.update(pkt, gamma)
    te = np.sqrt(err2) / T
    te[dead] = np.nan
    return dict(tracking_error=te, not_optimal=not_optimal, infeasible=dead, x_final=x)


def plant_callable(plant):
    """'cartpole' -> the numpy counterpart of the device plant (workloads.cartpole_step); callables pass through."""
    if callable(plant):
        return plant
    if plant == "cartpole":
        from . import workloads

        def step(x, u):
            return workloads.cartpole_step(x, u[:, 0])
        step.trace = lambda x, u: workloads.cartpole_trace(x, u[:, 0])       # physics-rate states (tracking error at 500 Hz)
        return step
    raise ValueError(f"unknown plant {plant!r}")


def mc_sweep(mpc, model: dict, p_loss, n_mc: int, T: int, ref, seed: int = 20240301, rank: int = 0, world: int = 1,
             extended: bool = False, device=None, on_device: bool = False, plant=None, warm_start: bool = False,
             timing: bool = False, device_rng: bool = False, force_collective: bool = False):
    """The Monte-Carlo sweep of results_linear_system.py:147-301 (BASELINE config 4): len(p_loss) x n_mc
    trajectories of T steps, sharded over `world` ranks (one process per GPU, contiguous p_loss-balanced
    shards), every time step of a shard solved by one kernel launch, statistics all-gathered at the end.
    Returns (table (n_total, 3) = [tracking error, tube violations, non-optimal solves], p_index (n_total,)),
    identical on every rank.  timing (device loop only): two more columns, the mean and the maximum device time of a
    trajectory's solves in seconds -- the computational times results_linear_system.py:305-315 reports.
    device_rng: Philox streams keyed by (seed, global trajectory index), drawn on the device in the device loop
    (tmpc_mc_set_device_rng) and by draw_realisations_philox, the same numbers, in the host loops."""
    import torch
    p_loss = np.asarray(p_loss, dtype=np.float64)
    pi, _ = trajectory_table(p_loss, n_mc)
    n_total = len(pi)
    lo, hi = shard_bounds(n_total, rank, world)
    if device_rng and on_device:
        th = ga = w = None
    elif device_rng:
        th, ga, w = draw_realisations_philox(hi - lo, T, model["w_bound"], seed=seed, first=lo)
    else:
        th, ga, w = draw_realisations(hi - lo, T, model["w_bound"], seed=seed, first=lo)
    ref = np.broadcast_to(np.asarray(ref, dtype=np.float64), (T,))
    if on_device:        # state machines on the GPU as well (tmpc_mc_run); otherwise the host loop around determine_packets
        out = mpc.run_closed_loop(p_loss[pi[lo:hi]], ref, th, ga, w, extended=extended, plant=plant, warm_start=warm_start,
                                  timing=timing, device_rng=(seed, lo, model["w_bound"]) if device_rng else None)
    elif getattr(mpc, "_smart_actuator", False):       # TrackingMPC: the comparator's loop (results_linear_system.py:262-287)
        out = run_remote_tracking_mpc(mpc.determine_packets, model["A"], model["B"], mpc.get_steady_state_controller_gain(), mpc._N,
                                      p_loss[pi[lo:hi]], ref, th, ga, w)
        out["tube_violations"] = np.zeros(hi - lo, dtype=np.int32)
    else:
        out = run_remote_tube_mpc(mpc.determine_packets, model["A"], model["B"], mpc.get_steady_state_controller_gain(),
                                  mpc.get_ancillary_controller_gain(), mpc._N, mpc._Z, p_loss[pi[lo:hi]], ref, th, ga, w,
                                  extended=extended, plant=None if plant is None else plant_callable(plant))
    cols = [out["tracking_error"], out["tube_violations"], out["not_optimal"]]
    if timing and on_device:
        cols += [out["solve_time_mean"], out["solve_time_max"]]
    local = torch.tensor(np.column_stack(cols), dtype=torch.float64)
    if device is not None:
        local = local.to(device)
    table = gather_statistics(local, n_total, rank, world, force_collective=force_collective)
    return table.cpu().numpy(), pi
