"""Observation-sharded BA across the GPUs of one node (one process per GPU, RCCL over xGMI).

The observation rows of ONE window are split into contiguous slices, one per rank (rows are frame sorted,
reference ``estimation/od_pipe.py:219-228``).  The stages that are indexed by observation (reprojection,
robust weights, per-pose accumulation, trial residuals) run on the local slice; the pose-chain stages
(dynamics factor, assembly, block-tridiagonal solve, retraction) are tiny and run redundantly on every
rank.  Per ``BA()`` call three device buffers are exchanged with all-gathers and reduced in rank order, so
every rank holds bit-identical normal equations and takes the same LM decisions:

1. ``|r|`` keys (16 B per observation) -> exact global lower median (``BA_filtering.py:23``);
2. per-pose partial sums of ``w J^T J`` / ``w J^T r`` (27 doubles per pose -- the "reduced camera" normal
   equations) plus the local max weight and ``sum |r|``;
3. two doubles per LM trial (weighted trial residual sums) for the accept test (``BA_filtering.py:73``).

The class is written against a small stage interface so that the same control flow is exercised on CPU
tensors with the gloo backend in the test-suite (with a stand-in engine) and on the GPU with RCCL.
"""
from __future__ import annotations

import math
import os

import numpy as np


def shard_bounds(m_total: int, world: int):
    """Contiguous, near-equal row slices: rank r owns [b[r], b[r+1])."""
    base, rem = divmod(m_total, world)
    sizes = [base + (1 if r < rem else 0) for r in range(world)]
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


def loaded_rccl_path():
    """The RCCL shared object this process has mapped already (torch's own copy once torch is imported), else ROCm's."""
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.split(None, 5)[-1].strip()
                if "librccl" in os.path.basename(path):
                    return path
    except OSError:
        pass
    return "/opt/rocm/lib/librccl.so"


class HipStageEngine:
    """Stage interface backed by libvinsat_ba.so; buffers are torch CUDA tensors, exchanged by pointer.

    ``attach_rccl`` hands the exchanges to the library itself (``vba_sh_comm_init`` / ``vba_sh_call``): RCCL all-gathers
    on the handle's stream between the stage kernels, one host call per ``BA()`` call."""

    native = False

    def __init__(self, eng, torch_stream=True):
        import torch
        self.eng = eng
        self.lib = eng.lib
        self.torch = torch
        from ._lib import check
        self._check = check
        self._torch_stream = bool(torch_stream)
        if torch_stream:
            check(self.lib.vba_set_stream(eng.h, torch.cuda.current_stream().cuda_stream, 1), self.lib)

    def attach_rccl(self, dist, group=None, rccl_path=None):
        """Join the library-owned communicator of this window's ranks.  ``dist`` / ``group``: any control plane that can
        ``broadcast_object_list`` (a gloo group will do: only the 128-byte id travels over it).  Collective."""
        import ctypes
        path = (rccl_path or loaded_rccl_path()).encode()
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        buf = ctypes.create_string_buffer(128)
        if rank == 0:
            self._check(self.lib.vba_sh_unique_id(path, buf), self.lib)
        box = [bytes(buf.raw)]
        src = dist.get_global_rank(group, 0) if group is not None and hasattr(dist, "get_global_rank") else 0
        dist.broadcast_object_list(box, src=src, group=group)
        self._check(self.lib.vba_sh_comm_init(self.eng.h, path, box[0], int(world), int(rank)), self.lib)
        self.native = True
        self.rccl_path = path.decode()

    def call(self, it, init, m_total):
        """One BA() call through ``vba_sh_call``; returns the number of rounds of the LM loop."""
        import ctypes
        ntr = ctypes.c_int()
        self._check(self.lib.vba_sh_call(self.eng.h, int(it), int(bool(init)), int(m_total), ctypes.byref(ntr)), self.lib)
        return ntr.value

    def run_schedule(self, iters, inits, m_total):
        """``len(iters)`` consecutive BA() calls as ONE host call, chained on the device (``vba_sh_run_schedule``)."""
        import ctypes
        n = len(iters)
        a = (ctypes.c_int * n)(*[int(x) for x in iters])
        b = (ctypes.c_int * n)(*[int(bool(x)) for x in inits])
        t = ctypes.c_int()
        self._check(self.lib.vba_sh_run_schedule(self.eng.h, n, a, b, int(m_total), ctypes.byref(t)), self.lib)
        return t.value

    def set_protocol(self, carried_keys):
        """True (default): the carried-keys protocol; False: every call gathers all |r| keys (round 3)."""
        self._check(self.lib.vba_sh_set_protocol(self.eng.h, int(bool(carried_keys))), self.lib)

    def stats(self):
        """(bytes of this rank's first exchange per call, calls repeated after a missed select, calls finished by the LM loop)."""
        import ctypes
        a, b, c = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        self._check(self.lib.vba_sh_stats(self.eng.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)), self.lib)
        return a.value, b.value, c.value

    def partial_count(self, n):
        return int(self.lib.vba_sh_partial_count(int(n)))

    def new_buffer(self, count):
        return self.torch.empty(int(count), dtype=self.torch.float64, device="cuda")

    def stage1(self, it, init, m_total, abs_local):
        self._check(self.lib.vba_sh_stage1(self.eng.h, int(it), int(bool(init)), int(m_total), abs_local.data_ptr()), self.lib)

    def stage2(self, abs_all, partial_local):
        self._check(self.lib.vba_sh_stage2(self.eng.h, abs_all.data_ptr(), abs_all.numel(), partial_local.data_ptr()), self.lib)

    def stage3(self, partial_all, ranks, trial_local):
        ptr = partial_all.data_ptr() if partial_all is not None else None
        self._check(self.lib.vba_sh_stage3(self.eng.h, ptr, int(ranks), trial_local.data_ptr()), self.lib)

    def stage4(self, trial_all, ranks):
        import ctypes
        done = ctypes.c_int()
        self._check(self.lib.vba_sh_stage4(self.eng.h, trial_all.data_ptr(), int(ranks), ctypes.byref(done)), self.lib)
        return bool(done.value)

    def set_states(self, states, lamda):
        self.eng.set_states(states, lamda)

    def get_states(self):
        return self.eng.get_states()

    def close(self):
        if self.native:
            self.lib.vba_sh_comm_destroy(self.eng.h)
        if self._torch_stream:
            self.lib.vba_set_stream(self.eng.h, None, 0)
        self.eng.close()


class HostStagedCollectives:
    """``all_gather_into_tensor`` for DEVICE tensors over a CPU process group (gloo): device -> host, gather, host -> device.

    A rehearsal / test transport, not a fast path: RCCL needs one device per rank, so two ranks that share a GPU -- the only
    way to run the multi-process + HIP-stage-kernel combination on a one-GPU box -- cannot use it.  Same interface as the
    ``torch.distributed`` module as far as :class:`ShardedBA` uses it."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group

    def get_world_size(self, group=None):
        return self.dist.get_world_size(self.group)

    def get_rank(self, group=None):
        return self.dist.get_rank(self.group)

    def all_gather_into_tensor(self, out, inp, group=None):
        host_in = inp.detach().cpu()                    # (synchronises with the stream the stage kernels ran on)
        host_out = self.torch.empty(out.shape, dtype=out.dtype)
        self.dist.all_gather_into_tensor(host_out, host_in, group=self.group)
        out.copy_(host_out)


class ShardedBA:
    """One window, rows sharded over the ranks of ``group`` (default: the world group)."""

    def __init__(self, engine, n, m_local, m_total, group=None, collectives=None):
        import torch.distributed as dist
        if collectives is not None:
            dist = collectives
        self.dist = dist
        self.group = group
        self.engine = engine
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n, self.m_local, self.m_total = int(n), int(m_local), int(m_total)
        self.m_pad = int(math.ceil(m_total / self.world))        # equal-size all-gather slots
        self.n_trials = 0
        e = engine
        if getattr(e, "native", False):     # the library owns the exchange buffers and issues the collectives itself
            return
        self.abs_local = e.new_buffer(2 * self.m_pad)
        self.abs_local.fill_(float("inf"))                       # padding sorts above every |r|
        self.abs_all = e.new_buffer(2 * self.m_pad * self.world)
        pc = e.partial_count(n)
        self.partial_local = e.new_buffer(pc)
        self.partial_all = e.new_buffer(pc * self.world)
        self.trial_local = e.new_buffer(2)
        self.trial_all = e.new_buffer(2 * self.world)
        self.n_trials = 0

    @classmethod
    def from_window(cls, win, device=0, group=None, collectives=None, native=False, rccl_path=None):
        """Build the GPU-backed sharded solver for a :class:`vinsat_amd.od_pipe.Window` on this rank.

        ``native``: the library issues the all-gathers itself (RCCL on its own stream, ``vba_sh_call``); ``group`` is then
        only the control plane over which the communicator's id is handed out (any backend); ``rccl_path`` names the library
        that provides the collectives (default: the RCCL this process has loaded)."""
        import torch.distributed as dist
        from .engine import BAEngine
        if collectives is not None:
            dist = collectives
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        n, m = win.time_idx.size, win.ii.size
        b = shard_bounds(m, world)
        lo, hi = int(b[rank]), int(b[rank + 1])
        # every rank sizes its handle for ceil(m / world) rows: the exchange buffers of the library-issued protocol are laid out by
        # the handle's geometry (observation blocks, bucket capacity) and must be the same on all ranks
        eng = BAEngine(n, max(-(-m // world), 1), windows=1, device=device)
        eng.upload_observations(win.landmarks_xyz[lo:hi], win.landmarks_uv[lo:hi], win.confidences[lo:hi], win.ii[lo:hi], n)
        eng.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
        if native:
            stage = HipStageEngine(eng, torch_stream=False)
            stage.attach_rccl(dist, group, rccl_path)
            return cls(stage, n, hi - lo, m, group, collectives)
        return cls(HipStageEngine(eng), n, hi - lo, m, group, collectives)

    def set_states(self, states, lamda):
        self.engine.set_states(states, lamda)

    def get_states(self):
        return self.engine.get_states()

    def step(self, it, initialize):
        """One ``BA()`` call; returns the number of LM trials."""
        d, e = self.dist, self.engine
        if getattr(e, "native", False):
            self.n_trials = e.call(it, initialize, self.m_total)
            return self.n_trials
        e.stage1(it, initialize, self.m_total, self.abs_local)
        d.all_gather_into_tensor(self.abs_all, self.abs_local, group=self.group)
        e.stage2(self.abs_all, self.partial_local)
        d.all_gather_into_tensor(self.partial_all, self.partial_local, group=self.group)
        first = True
        trials = 0
        while True:
            e.stage3(self.partial_all if first else None, self.world, self.trial_local)
            d.all_gather_into_tensor(self.trial_all, self.trial_local, group=self.group)
            trials += 1
            first = False
            if e.stage4(self.trial_all, self.world):
                break
            if trials >= 24:        # lamda runs out after 9 trials (+ one repeat for a pivoted fallback): the device never reported an outcome
                raise RuntimeError(f"sharded BA call (iter {it}): LM loop did not terminate within {trials} trials")
        self.n_trials = trials
        return trials

    def run_schedule(self, iters, inits):
        """``len(iters)`` consecutive ``BA()`` calls; with the library-issued exchanges one host call, chained on the device."""
        e = self.engine
        if getattr(e, "native", False):
            self.n_trials = e.run_schedule(iters, inits, self.m_total)
            return self.n_trials
        total = 0
        for it, init in zip(iters, inits):
            total += self.step(it, init)
        return total

    def close(self):
        self.engine.close()
