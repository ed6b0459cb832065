"""Stand-in stage engine for vinsat_amd.dist.ShardedBA built on the CPU oracle (test infrastructure).

It implements the four-stage protocol of include/vinsat_ba.h (vba_sh_stage1..4) on CPU torch tensors so that the
sharding / padding / all-gather / LM-loop control flow of ShardedBA can run under the gloo backend without a GPU.
"""
import numpy as np
import torch

from oracle import ba_oracle as O


class OracleStageEngine:
    def __init__(self, xyz, uv, conf, ii, K, cumrot, time_idx):
        self.xyz, self.uv, self.conf, self.ii = xyz, uv, conf, np.asarray(ii, dtype=np.int64)
        self.K, self.cumrot, self.t = K, cumrot, np.asarray(time_idx, dtype=np.int64)
        self.n = K.shape[0]
        self.states = None
        self.lam = None
        self.hess = np.zeros((9, 9))
        self.n_trials = 0
        self.flags = 0

    # ---- interface used by ShardedBA
    def partial_count(self, n):
        return 27 * n + 2

    def new_buffer(self, count):
        return torch.zeros(int(count), dtype=torch.float64)

    def set_states(self, states, lamda):
        self.states, self.lam = np.array(states, dtype=np.float64), float(lamda)

    def get_states(self):
        return self.states.copy(), self.lam, self.hess.copy(), self.n_trials, self.flags

    def stage1(self, it, init, m_total, abs_local):
        self.it, self.init, self.m_total = it, bool(init), int(m_total)
        self.alpha, self.sigma = O.lm_schedule(it)
        self.n_trials, self.flags = 0, 0
        est, self.Jg = O.landmark_project(self.states, self.xyz, self.K, self.ii, jacobian=True)
        self.r_obs = self.uv - est
        a = np.abs(self.r_obs).reshape(-1)
        abs_local[: a.size] = torch.from_numpy(a)
        self.sum_abs = a.sum()

    def stage2(self, abs_all, partial_local):
        keys = np.sort(abs_all.numpy())
        c = keys[(2 * self.m_total - 1) // 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            per = (((self.r_obs / c) ** 2) / abs(self.alpha - 2) + 1) ** (self.alpha / 2 - 1) / (c ** 2)
        self.w_raw = per.mean(-1)
        H, b = O.accumulate(self.Jg, self.w_raw * self.conf, self.r_obs, self.ii, self.n)
        iu = np.triu_indices(6)
        out = np.concatenate([H[:, iu[0], iu[1]].reshape(-1), b.reshape(-1), [self.w_raw.max() if self.w_raw.size else 0.0, self.sum_abs]])
        partial_local[:] = torch.from_numpy(out)

    def stage3(self, partial_all, ranks, trial_local):
        n = self.n
        if partial_all is not None:
            P = partial_all.numpy().reshape(ranks, 27 * n + 2)
            tot = np.zeros(27 * n)
            for q in range(ranks):
                tot = tot + P[q, : 27 * n]
            self.wmax = P[:, 27 * n].max()
            sum_abs = 0.0
            for q in range(ranks):
                sum_abs += P[q, 27 * n + 1]
            Hp = tot[: 21 * n].reshape(n, 21)
            iu = np.triu_indices(6)
            H = np.zeros((n, 6, 6))
            H[:, iu[0], iu[1]] = Hp
            H[:, iu[1], iu[0]] = Hp
            b = tot[21 * n:].reshape(n, 6)
            if self.init:
                r_pred = np.zeros((n - 1, 6))
                E = F = r_orb = qg = Hd = Hu = Hl = None
            else:
                r_orb, E, F = O.orbit_factor(self.states, self.t, jacobian=True)
                f, qg, Hd, Hu, Hl = O.attitude_factor(self.states, self.cumrot, jacobian=True)
                r_pred = np.concatenate([r_orb, f[:, None]], -1)
            self.bands, self.rhs = O.assemble(H, b, 1.0 / self.wmax, float(self.sigma), E, F, r_orb, qg, Hd, Hu, Hl, self.init)
            self.denom = 2.0 * self.m_total + r_pred.shape[1] * (n - 1)
            self.init_residual = (sum_abs + np.abs(r_pred).sum() * np.sqrt(self.sigma)) / self.denom
        self.lam32 = float(np.float32(self.lam))
        A = self.bands.copy()
        A[:, 1] += self.lam32 * np.eye(9)
        self.A_last = A
        dpose = O.solve_tridiag(A, self.rhs)
        self.states_new = O.retract(self.states, dpose)
        est1 = O.landmark_project(self.states_new, self.xyz, self.K, self.ii, jacobian=False)
        w = (self.w_raw / self.wmax) * self.conf
        so = np.abs((self.uv - est1) * w[:, None]).sum()
        sd = np.abs(O.dynamics_residual(self.states_new, self.cumrot, self.t, self.init)).sum() * np.sqrt(self.sigma)
        trial_local[0], trial_local[1] = float(so), float(sd)

    def stage4(self, trial_all, ranks):
        t = trial_all.numpy()
        S = t[1]
        for q in range(ranks):
            S += t[2 * q]
        residual = S / self.denom
        lam = self.lam * 10
        self.n_trials += 1
        accept = residual < self.init_residual
        if accept or lam > 1e4:
            if not accept:
                self.flags |= 1
            self.lam = max(min(1e-1, lam * 0.01), 1e-4)
            self.states = self.states_new
            self.hess = self.A_last[-1, 1].copy()
            return True
        self.lam = lam
        return False

    def close(self):
        pass
