#!/usr/bin/env python3
"""Generate the golden parity fixtures under tests/golden/ by RUNNING the reference.

Runs only in the build container (needs /root/reference; never runs on the GPU box).
The reference's own driver ``estimation/od_pipe.py:streaming_version`` is imported
unmodified and fed the synthetic sequences of ``vinsat_amd.synth``; every call it makes
to ``BA`` (``estimation/BA/BA_filtering.py:4-98``) is intercepted to record inputs,
outputs and -- for selected iterations -- the intermediates (projection, Jacobian,
dynamics factor, attitude Newton term, the matrix and right-hand side handed to
``torch.linalg.solve`` and its solution for every LM trial).

Two third-party modules the reference imports are not installed in this image and are
replaced by minimal equivalents before the import:
``torch_scatter`` (segment sum / mean, semantics fixed by the call sites
``BA_utils.py:1376-1382``) and ``ipdb`` (debugger hook, no-op).

Usage: python tools/gen_golden.py [C1 C2 C3 C4 GAP HOP REGC1 REGC2 PRIORPROP REJ C5S HOPC2 HOPGAP]
Outputs are data only (inputs + expected outputs), compressed .npz.
"""
from __future__ import annotations

import os
import sys
import time
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/estimation"
sys.path.insert(0, REPO)

from vinsat_amd import synth  # noqa: E402


def _install_stubs():
    ts = types.ModuleType("torch_scatter")

    def scatter_sum(src, index, dim=-1, dim_size=None):
        shape = list(src.shape)
        shape[dim] = dim_size
        out = torch.zeros(shape, dtype=src.dtype, device=src.device)
        return out.index_add_(dim, index, src)

    def scatter_mean(src, index, dim=-1, dim_size=None):
        s = scatter_sum(src, index, dim, dim_size)
        cnt = torch.zeros(dim_size, dtype=src.dtype).index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
        cnt = cnt.clamp(min=1)
        view = [1] * s.dim()
        view[dim] = dim_size
        return s / cnt.view(view)

    ts.scatter_sum = scatter_sum
    ts.scatter_mean = scatter_mean
    sys.modules["torch_scatter"] = ts
    ip = types.ModuleType("ipdb")
    ip.set_trace = lambda *a, **k: None
    sys.modules["ipdb"] = ip


def load_reference():
    _install_stubs()
    sys.path.insert(0, REF)
    os.chdir(REF)
    import matplotlib
    matplotlib.use("Agg")
    import od_pipe  # noqa
    import BA.BA_filtering as baf  # noqa
    return od_pipe, baf


def tridiag_bands(A, n):
    """Dense [9n,9n] -> bands [n,3,9,9] (sub, diag, super) + max |entry| outside the band."""
    A = A.reshape(n, 9, n, 9).transpose(0, 2, 1, 3)  # [i,j,9,9]
    bands = np.zeros((n, 3, 9, 9))
    mask = np.ones((n, n), bool)
    for i in range(n):
        for d, j in enumerate((i - 1, i, i + 1)):
            if 0 <= j < n:
                bands[i, d] = A[i, j]
                mask[i, j] = False
    off = np.abs(A[mask]).max() if mask.any() else 0.0
    return bands, float(off)


class Capture:
    def __init__(self, od_pipe, baf, full_iters):
        self.od_pipe, self.baf = od_pipe, baf
        self.full_iters = set(full_iters)
        self.calls = []
        self.cur = None
        self.timing = []

    def install(self):
        baf, od_pipe = self.baf, self.od_pipe
        self._BA = baf.BA
        self._lp = baf.landmark_project
        self._pred = baf.predict
        self._solve = torch.linalg.solve
        cap = self

        def lp(poses, xyz, intr, ii, jacobian=True):
            out = cap._lp(poses, xyz, intr, ii, jacobian=jacobian)
            if cap.cur is not None and cap.cur["full"]:
                if jacobian:
                    cap.cur["landmark_est"] = out[0].detach().numpy().copy()
                    cap.cur["Jg"] = out[1].detach().numpy().copy()
                else:
                    cap.cur.setdefault("trial_est", []).append(out.detach().numpy().copy())
            return out

        def pred(states, imu, times, qc, vc, dt=1, jacobian=True, initialize=False):
            out = cap._pred(states, imu, times, qc, vc, dt=dt, jacobian=jacobian, initialize=initialize)
            c = cap.cur
            if c is not None and c["full"]:
                n = states.shape[1]
                if jacobian:
                    c["r_pred"] = out[0].detach().numpy().copy()
                    if not initialize:
                        Jf = out[5].detach().numpy()[0].reshape(n - 1, 6, n, 9)
                        # keep only the two non-zero 6x9 blocks per row-block, assert the rest is 0
                        blk = np.zeros((n - 1, 2, 6, 9))
                        m = np.ones((n - 1, n), bool)
                        for i in range(n - 1):
                            blk[i, 0] = Jf[i, :, i]
                            blk[i, 1] = Jf[i, :, i + 1]
                            m[i, i] = m[i, i + 1] = False
                        c["Jf_blocks"] = blk
                        c["Jf_offband"] = float(np.abs(Jf.transpose(0, 2, 1, 3)[m]).max()) if m.any() else 0.0
                        Hq, off = tridiag_bands(out[6].detach().numpy()[0], n)
                        c["Hq_bands"], c["Hq_offband"] = Hq, off
                        c["qgrad"] = out[7].detach().numpy().copy()
                else:
                    c.setdefault("trial_r_pred", []).append(out[0].detach().numpy().copy())
            return out

        def solve(A, b, *a, **k):
            x = cap._solve(A, b, *a, **k)
            c = cap.cur
            if c is not None:
                c["n_trials"] += 1
                if c["full"]:
                    n = A.shape[-1] // 9
                    bands, off = tridiag_bands(A.detach().numpy()[0], n)
                    c.setdefault("A_bands", []).append(bands)
                    c.setdefault("A_offband", []).append(off)
                    c.setdefault("JTr", []).append(b.detach().numpy().copy())
                    c.setdefault("dpose", []).append(x.detach().numpy().copy())
            return x

        def BA(iter, states, velocities, imu_meas, landmarks, landmarks_xyz, ii, time_idx, intrinsics,
               confidences, Sigma, V, lamda_init, poses_gt_eci, initialize=False):
            k = len(cap.calls)
            c = {"call": k, "iter": iter, "initialize": bool(initialize), "full": k in cap.full_iters,
                 "n_trials": 0, "lamda_in": float(lamda_init),
                 "states_in": states.detach().double().numpy().copy()}
            if k == 0:
                cap.inputs = dict(
                    velocities=velocities.detach().double().numpy().copy(),
                    cumrot_last=imu_meas.detach().double().numpy()[0, :, -1, 6:10].copy(),
                    imu_shape=np.array(imu_meas.shape),
                    landmarks=landmarks.detach().double().numpy().copy(),
                    landmarks_xyz=landmarks_xyz.detach().double().numpy().copy(),
                    ii=np.asarray(ii).astype(np.int64).copy(),
                    time_idx=np.asarray(time_idx).astype(np.int64).copy(),
                    intrinsics=intrinsics.detach().double().numpy().copy(),
                    confidences=confidences.detach().double().numpy().copy(),
                    poses_gt_eci=poses_gt_eci.detach().double().numpy().copy(),
                )
            cap.cur = c
            t0 = time.perf_counter()
            out = cap._BA(iter, states, velocities, imu_meas, landmarks, landmarks_xyz, ii, time_idx, intrinsics,
                          confidences, Sigma, V, lamda_init, poses_gt_eci, initialize=initialize)
            c["seconds"] = time.perf_counter() - t0
            cap.cur = None
            c["states_out"] = out[0].detach().numpy().copy()
            c["lamda_out"] = float(out[2])
            c["last_hessian"] = out[3].detach().numpy().copy()
            cap.calls.append(c)
            return out

        baf.landmark_project = lp
        baf.predict = pred
        torch.linalg.solve = solve
        od_pipe.BA = BA

    def uninstall(self):
        self.baf.landmark_project = self._lp
        self.baf.predict = self._pred
        torch.linalg.solve = self._solve
        self.od_pipe.BA = self._BA


def run_config(name, od_pipe, baf, full_iters, store_inputs, store_states, drop=(), hop=False):
    if hop:
        # The reference's GPU default: on a machine with a GPU, BA takes predict_gpu (BA_filtering.py:16-17), which is
        # predict (BA_utils.py:457-527) with the coarse integrator propagate_orbit_dynamics_skip (:52-71) in place of
        # propagate_orbit_dynamics (:73-87) and nothing else changed (the two functions differ in that line and in .cuda()
        # / .cpu() moves only).  Here: the CPU branch with that one function swapped, i.e. predict_gpu's arithmetic.
        import BA.BA_utils as bu
        orig_prop = bu.propagate_orbit_dynamics
        bu.propagate_orbit_dynamics = bu.propagate_orbit_dynamics_skip
        try:
            return run_config(name, od_pipe, baf, full_iters, store_inputs, store_states, drop)
        finally:
            bu.propagate_orbit_dynamics = orig_prop
    if name in ("GAP", "HOPGAP"):
        det, orbit = synth.make_two_pass_sequence()
    elif name == "HOPC2":
        det, orbit = synth.make_sequence("C2", seed=0)
    elif name == "REJ":
        # confidences of 3 (> 1): the weighted trial residual (BA_filtering.py:66-69) no longer undercuts the unweighted
        # initial one for free, so the LM loop of plain BA rejects trials (:72-77) -- 1 to 9 trials per call
        det, orbit = synth.make_sequence("C2", seed=3, conf=3.0)
    elif name == "C5S":
        det, orbit = synth.make_subwindow("C5", 500)
    else:
        det, orbit = synth.make_sequence(name, seed=0)
    cap = Capture(od_pipe, baf, full_iters)
    cap.install()
    try:
        t0 = time.perf_counter()
        errors, first_det, times = od_pipe.streaming_version(detections=det.copy(), orbit_np=orbit.copy())
        wall = time.perf_counter() - t0
    finally:
        cap.uninstall()
    out = {}
    inp = cap.inputs
    if store_inputs:
        for k, v in inp.items():
            out["in_" + k] = v
    else:
        # digest only: enough to check that the regenerated inputs are the same arrays
        for k in ("landmarks", "landmarks_xyz", "confidences", "intrinsics", "cumrot_last"):
            v = inp[k].reshape(-1)
            out["digest_" + k] = np.array([v.size, v.sum(), np.abs(v).sum(), v[0], v[v.size // 2], v[-1]])
        out["in_ii_digest"] = np.array([inp["ii"].size, inp["ii"].sum(), inp["ii"][0], inp["ii"][-1]])
        out["in_time_idx"] = inp["time_idx"]
        out["in_poses_gt_eci"] = inp["poses_gt_eci"]
        out["in_velocities"] = inp["velocities"]
    ncall = len(cap.calls)
    out["n_poses_per_call"] = np.array([c["states_in"].shape[1] for c in cap.calls])
    out["iters"] = np.array([c["iter"] for c in cap.calls])
    out["initialize"] = np.array([c["initialize"] for c in cap.calls])
    out["n_trials"] = np.array([c["n_trials"] for c in cap.calls])
    out["lamda_in"] = np.array([c["lamda_in"] for c in cap.calls])
    out["lamda_out"] = np.array([c["lamda_out"] for c in cap.calls])
    out["seconds"] = np.array([c["seconds"] for c in cap.calls])
    out["states0"] = cap.calls[0]["states_in"]
    keep = range(ncall) if store_states == "all" else store_states
    for k in keep:
        out[f"states_out_{k}"] = cap.calls[k]["states_out"]
        out[f"last_hessian_{k}"] = cap.calls[k]["last_hessian"]
    for c in cap.calls:
        if not c["full"]:
            continue
        k = c["call"]
        out[f"states_in_{k}"] = c["states_in"]
        for key in ("landmark_est", "Jg", "r_pred", "Jf_blocks", "Hq_bands", "qgrad"):
            if key in c and key not in drop:
                out[f"{key}_{k}"] = c[key]
        for key in ("Jf_offband", "Hq_offband"):
            if key in c:
                out[f"{key}_{k}"] = np.array(c[key])
        out[f"A_bands_{k}"] = np.stack(c["A_bands"])
        out[f"A_offband_{k}"] = np.array(c["A_offband"])
        out[f"JTr_{k}"] = np.stack(c["JTr"])
        out[f"dpose_{k}"] = np.stack(c["dpose"])
        if "trial_est" not in drop:
            out[f"trial_est_{k}"] = np.stack(c["trial_est"])
        out[f"trial_r_pred_{k}"] = np.stack(c["trial_r_pred"])
    out["errors"] = errors.detach().numpy()
    out["first_detection"] = np.array(first_det)
    out["times"] = np.concatenate([np.atleast_1d(np.asarray(t)) for t in times])
    out["ref_wall_seconds"] = np.array(wall)
    out["ref_threads"] = np.array(torch.get_num_threads())
    path = os.path.join(REPO, "tests", "golden", f"{name.lower()}.npz")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, **out)
    secs = out["seconds"]
    print(f"[{name}] calls={ncall} n_trials={out['n_trials'].tolist()} wall={wall:.1f}s "
          f"init-phase {secs[out['initialize']].mean():.3f}s/it full-phase {secs[~out['initialize']].mean():.3f}s/it "
          f"final err {float(errors[-1]):.4f} km -> {path} ({os.path.getsize(path)/1e6:.2f} MB)")


def run_hop(od_pipe, baf):
    """Golden vectors of the coarse integrator: outputs of the reference's propagate_orbit_dynamics_skip
    (BA_utils.py:52-71) and its autograd Jacobian, for gaps below, at and above the 100 s hop."""
    import BA.BA_utils as bu
    gaps = np.array([3, 100, 250, 99, 101, 1, 400, 37, 200, 945, 555])
    times = np.concatenate([[10], 10 + np.cumsum(gaps)]).astype(np.int64)
    traj = synth.integrate_orbit(int(times[-1]) + 2)
    rng = np.random.default_rng(7)
    x = traj[times] + np.concatenate([rng.normal(0, 5.0, (len(times), 3)), rng.normal(0, 0.01, (len(times), 3))], 1)
    xt = torch.tensor(x)

    def f(xx):
        p, v = bu.propagate_orbit_dynamics_skip(xx[None, :, :3], xx[None, :, 3:], times, 1)
        return torch.cat([p[0], v[0]], -1)

    out = f(xt)
    J = torch.autograd.functional.jacobian(f, xt)            # [n,6,n,6]
    n = len(times)
    Phi = np.stack([J[i, :, i, :].numpy() for i in range(n)])
    off = J.clone()
    for i in range(n):
        off[i, :, i, :] = 0
    path = os.path.join(REPO, "tests", "golden", "hop.npz")
    np.savez_compressed(path, times=times, x=x, x_pred=out.numpy(), Phi=Phi, offdiag_max=float(off.abs().max()))
    print(f"[HOP] n={n} gaps={gaps.tolist()} -> {path}")


def run_reg(od_pipe, baf, base="c1", calls=range(20), full=(10, 11, 19)):
    """Golden vectors of ``BA_reg`` (BA_filtering.py:100-210), the BA call with a propagated-covariance prior.

    Its only caller in the reference cannot run, so the inputs are made here: the window of fixture ``base``, a
    prior state per pose (the reference's own final estimate plus noise) and random symmetric positive definite
    prior information matrices.  ``prior_gpu`` (BA_utils.py:604-676) moves its arguments with ``.cuda()``
    unconditionally; on this CPU-only container that call is made the identity for the duration of the run
    (``torch.cuda.is_available()`` stays False, so ``BA_reg`` takes its ``predict`` branch).  ``ipdb`` is the
    no-op of the other fixtures."""
    g = np.load(os.path.join(REPO, "tests", "golden", f"{base}.npz"))
    n = g["states0"].shape[1]
    rng = np.random.default_rng(11)
    ref = g["states_out_19"][0]
    prior = ref.copy()
    prior[:, :3] += rng.normal(0, 0.05, (n, 3))
    prior[:, 7:] += rng.normal(0, 1e-4, (n, 3))
    dq = np.concatenate([rng.normal(0, 2e-3, (n, 3)), np.ones((n, 1))], 1)
    q = ref[:, 3:7]
    x, y, z, w = q.T
    dx, dy, dz, dw = dq.T
    prior[:, 3:7] = np.stack([w * dx + x * dw + y * dz - z * dy, w * dy - x * dz + y * dw + z * dx,
                              w * dz + x * dy - y * dx + z * dw, w * dw - x * dx - y * dy - z * dz], 1)
    prior[:, 3:7] /= np.linalg.norm(prior[:, 3:7], axis=1, keepdims=True)
    Hs = np.zeros((n, 6, 6))
    Hr = np.zeros((n, 3, 3))
    for i in range(n):
        R = rng.normal(0, 1, (6, 6))
        S = np.diag([2.0, 2.0, 2.0, 20.0, 20.0, 20.0])
        Hs[i] = S @ (np.eye(6) + 0.2 * R @ R.T / 6) @ S * rng.uniform(0.5, 1.5)
        R3 = rng.normal(0, 1, (3, 3))
        Hr[i] = 1e3 * (np.eye(3) + 0.3 * R3 @ R3.T)
    imu = np.zeros(tuple(g["in_imu_shape"]))
    imu[0, :, -1, 6:10] = g["in_cumrot_last"]
    t = lambda a: torch.tensor(np.asarray(a))
    args = dict(velocities=t(g["in_velocities"]), imu=t(imu), lm=t(g["in_landmarks"]), xyz=t(g["in_landmarks_xyz"]),
                intr=t(g["in_intrinsics"]), conf=t(g["in_confidences"]), gt=t(g["in_poses_gt_eci"]))
    ii, time_idx = g["in_ii"], g["in_time_idx"]
    states_prior, hs, hr = t(prior[None]), t(Hs[None]), t(Hr[None])
    out = dict(base=np.array(base), states_prior=prior[None], velocity_prior=prior[None, :, 7:], hessian_state_t=Hs[None],
               hessian_rot_t=Hr[None], states0=g["states0"])
    solve0 = torch.linalg.solve
    cuda0 = torch.Tensor.cuda
    rec = {}

    def solve(A, b, *a, **k):
        xx = solve0(A, b, *a, **k)
        rec["n_trials"] += 1
        if rec["full"]:
            nn = A.shape[-1] // 9
            bands, off = tridiag_bands(A.detach().numpy()[0], nn)
            rec["A"].append(bands)
            rec["off"].append(off)
            rec["JTr"].append(b.detach().numpy().copy())
            rec["dpose"].append(xx.detach().numpy().copy())
        return xx

    torch.linalg.solve = solve
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        states, lam = t(g["states0"]), 1e-4
        iters, inits, ntr, lam_in, lam_out, secs = [], [], [], [], [], []
        for k in calls:
            init = k < 10
            rec.update(n_trials=0, full=k in full, A=[], off=[], JTr=[], dpose=[])
            out[f"states_in_{k}"] = states.detach().numpy().copy()
            t0 = time.perf_counter()
            res = baf.BA_reg(k, states.clone(), args["velocities"], states_prior, states_prior[:, :, 7:], hs, hr, args["imu"],
                             args["lm"], args["xyz"], ii, time_idx, args["intr"], args["conf"], None, None, lam, args["gt"],
                             initialize=init)
            secs.append(time.perf_counter() - t0)
            iters.append(k); inits.append(init); ntr.append(rec["n_trials"]); lam_in.append(lam)
            states, lam = res[0].detach(), float(res[2])
            lam_out.append(lam)
            out[f"states_out_{k}"] = states.numpy().copy()
            out[f"last_hessian_{k}"] = res[3].detach().numpy().copy()
            if rec["full"]:
                out[f"A_bands_{k}"] = np.stack(rec["A"])
                out[f"A_offband_{k}"] = np.array(rec["off"])
                out[f"JTr_{k}"] = np.stack(rec["JTr"])
                out[f"dpose_{k}"] = np.stack(rec["dpose"])
    finally:
        torch.linalg.solve = solve0
        torch.Tensor.cuda = cuda0
    out.update(iters=np.array(iters), initialize=np.array(inits), n_trials=np.array(ntr), lamda_in=np.array(lam_in),
               lamda_out=np.array(lam_out), seconds=np.array(secs))
    path = os.path.join(REPO, "tests", "golden", f"reg_{base}.npz")
    np.savez_compressed(path, **out)
    print(f"[REG/{base}] calls={len(iters)} n_trials={ntr} lamda_out={lam_out} {sum(secs):.1f}s -> {path} "
          f"({os.path.getsize(path)/1e6:.2f} MB)")


def run_prior_prop(od_pipe, baf):
    """Golden vectors of ``propagate_dynamics_cov_init`` (BA_utils.py:227-248) for a batch of one: the last pose,
    carried velocity and last Hessian block of the C2 fixture pushed over a 7 s gap and a 40 s window."""
    import BA.BA_utils as bu
    g = np.load(os.path.join(REPO, "tests", "golden", "c2.npz"))
    state = g["states_out_19"][0, -1]
    vel = g["in_velocities"].reshape(-1, 3)[-1]
    hess = g["last_hessian_19"][0]
    tdiff, duration = 7, 40
    rng = np.random.default_rng(5)
    omega = rng.normal(0, 2e-3, (tdiff + duration, 3)) + np.array([0.0, 1.1e-3, 0.0])
    out = bu.propagate_dynamics_cov_init(torch.tensor(state[None]), torch.tensor(vel[None]), torch.tensor(hess[None]),
                                         torch.tensor(omega[None]), tdiff, duration, 1)
    path = os.path.join(REPO, "tests", "golden", "prior_prop.npz")
    np.savez_compressed(path, state=state, velocity=vel, hessian=hess, omega=omega, tdiff=tdiff, duration=duration,
                        states_t=out[0].detach().numpy(), velocities_t=out[1].detach().numpy(),
                        hessian_state_t=out[2].detach().numpy(), hessian_rot_t=out[3].detach().numpy())
    print(f"[PRIORPROP] shapes {[tuple(o.shape) for o in out]} -> {path}")


PLAN = {
    "C1": dict(full_iters=range(20), store_inputs=True, store_states="all"),
    "C2": dict(full_iters=(0, 9, 10, 19), store_inputs=True, store_states="all"),
    "C3": dict(full_iters=(), store_inputs=False, store_states=(0, 9, 10, 14, 19)),
    "C4": dict(full_iters=(), store_inputs=False, store_states=(0, 9, 10, 14, 19)),
    "GAP": dict(full_iters=(), store_inputs=True, store_states="all"),
    # plain BA with rejected trials: every LM trial's system / right-hand side / solution of calls with 2, 4, 6 and 9 trials
    "REJ": dict(full_iters=(1, 5, 9, 16), store_inputs=True, store_states="all", drop=("landmark_est", "Jg", "Jf_blocks", "Hq_bands", "trial_est")),
    # the first 500 poses of the C5 orbit (3 s stride, 250 observations per pose): SURVEY 8(c)(ii)
    "C5S": dict(full_iters=(), store_inputs=False, store_states=(0, 9, 10, 14, 19)),
    # chained runs with the integrator the reference itself takes when it sees a GPU (predict_gpu): the C2 window (5 s gaps:
    # one 5 s step instead of five 1 s steps) and the two-pass sequence (a gap of several hundred seconds: 100 s hops)
    "HOPC2": dict(full_iters=(10, 19), store_inputs=True, store_states="all", hop=True, drop=("landmark_est", "Jg", "Hq_bands", "trial_est")),
    "HOPGAP": dict(full_iters=(), store_inputs=True, store_states="all", hop=True),
}


def main():
    names = sys.argv[1:] or ["C1", "C2"]
    od_pipe, baf = load_reference()
    for name in names:
        if name == "HOP":
            run_hop(od_pipe, baf)
        elif name == "PRIORPROP":
            run_prior_prop(od_pipe, baf)
        elif name.startswith("REG"):
            run_reg(od_pipe, baf, base=(name[3:] or "C1").lower())
        else:
            run_config(name, od_pipe, baf, **PLAN[name])


if __name__ == "__main__":
    main()
