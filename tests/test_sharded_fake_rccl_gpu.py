"""The library-issued exchanges of the observation-sharded mode (vba_sh_comm_init / vba_sh_call) with TWO ranks.

RCCL wants one device per rank, so on a one-GPU box the real library can only ever run world size 1
(tests/test_sharded_native_gpu.py), where its all-gather is a device copy.  libvinsat_ba.so resolves the collectives at run time
from a library the caller names; tests/fake_rccl/ is a test double of those entry points over host shared memory.  With it two
rank PROCESSES on the one GPU execute vba_sh_call: slots of ceil(m / R) rows padded with +inf (m is not divisible by 2 here),
rank-ordered reduce across real peers, the communicator's id handed over a gloo group -- against the caller-dispatched protocol
bit for bit (same stage kernels, exchanges by the host-staged transport), the unsharded engine and the oracle; then the rows
are split the other way round on the same handles (one rank's shard shrinks: the padding must be laid out again)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE_DIR = os.path.join(ROOT, "tests", "fake_rccl")
FAKE = os.path.join(FAKE_DIR, "libfake_rccl.so")
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from conftest import golden_inputs, load_golden
    from oracle import ba_oracle as O
    from vinsat_amd.dist import HipStageEngine, HostStagedCollectives, ShardedBA, shard_bounds
    from vinsat_amd.engine import BAEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        g = load_golden("c2")
        inp = golden_inputs(g)
        n = inp["K"].shape[0]
        m = inp["xyz"].shape[0] - 3          # not divisible by the world size: unequal shards, +inf padded slots
        xyz, uv, conf, ii = inp["xyz"][:m], inp["uv"][:m], inp["conf"][:m].copy(), inp["ii"][:m]
        conf[:] = 3.0                        # the LM loop rejects trials: several rounds (and all-gathers) per call
        b = shard_bounds(m, world)
        lo, hi = int(b[rank]), int(b[rank + 1])

        def engine(lo, hi):
            e = BAEngine(n, max(hi - lo, -(-m // world)))      # (room for the other split of the rows, below)
            e.upload_observations(xyz[lo:hi], uv[lo:hi], conf[lo:hi], ii[lo:hi], n)
            e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
            return e

        stage = HipStageEngine(engine(lo, hi), torch_stream=False)
        stage.attach_rccl(dist, None, FAKE)                 # the 128-byte id travels over the gloo group
        assert stage.native and stage.rccl_path == FAKE
        a = ShardedBA(stage, n, hi - lo, m)
        c = ShardedBA(HipStageEngine(engine(lo, hi)), n, hi - lo, m, collectives=HostStagedCollectives())
        single = engine(0, m) if rank == 0 else None
        st, lam = g["states0"][0], 1e-4
        a.set_states(st, lam)
        c.set_states(st, lam)
        ref, lam_ref, sg, lam_g = st.copy(), lam, st.copy(), lam
        rounds = []
        for it, init in [(0, True), (1, True), (2, True), (5, True), (10, False), (11, False), (12, False), (13, False)]:
            na, nc = a.step(it, init), c.step(it, init)
            sa, sc = a.get_states(), c.get_states()
            assert na == nc and np.array_equal(sa[0], sc[0]) and sa[1] == sc[1] and sa[3] == sc[3], it      # same kernels, another dispatcher
            ref, lam_ref, _, ntr_ref = O.ba_iteration(it, ref, inp["cumrot"], uv, xyz, ii, inp["time_idx"], inp["K"], conf, lam_ref, initialize=init)
            assert sa[3] == ntr_ref and sa[1] == lam_ref and np.abs(sa[0] - ref).max() / np.abs(ref).max() < 1e-7, it
            if single is not None:
                sg, lam_g, _, ntr_g, _ = single.iterate(it, init, lam_g, sg)
                assert ntr_g == sa[3] and lam_g == sa[1] and np.abs(sa[0] - sg).max() / np.abs(sg).max() < 1e-9, it
            t = torch.from_numpy(sa[0].copy())
            lst = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(lst, t)
            assert all(torch.equal(lst[0], x) for x in lst)                # every rank holds the same bits
            rounds.append(na)
        if world != 2:                      # (the re-split leg below is written for two ranks)
            if rank == 0:
                np.save(os.path.join(tmp, "rounds.npy"), np.array(rounds + [-1]))
                single.close()
            a.close()
            c.engine.eng.close()
            return
        # The same handles, rows split the other way round (rank 0 one row fewer, rank 1 one more, same total): rank 0's slot
        # now ends one row earlier and what lies behind its keys must be +inf again, not the old shard's last keys.
        cut = int(b[1]) - 1
        lo2, hi2 = (0, cut) if rank == 0 else (cut, m)
        assert hi2 - lo2 <= -(-m // world)
        for sb in (a, c):
            sb.engine.eng.upload_observations(xyz[lo2:hi2], uv[lo2:hi2], conf[lo2:hi2], ii[lo2:hi2], n)
        a.m_local = hi2 - lo2
        c2 = ShardedBA(c.engine, n, hi2 - lo2, m, collectives=HostStagedCollectives())      # (fresh, +inf padded exchange buffers)
        a.set_states(st, lam)
        c2.set_states(st, lam)
        ref2, lam2 = st.copy(), lam
        for it, init in [(0, True), (1, True), (10, False)]:
            na, nc = a.step(it, init), c2.step(it, init)
            sa, sc = a.get_states(), c2.get_states()
            assert na == nc and np.array_equal(sa[0], sc[0]) and sa[1] == sc[1], it
            ref2, lam2, _, ntr2 = O.ba_iteration(it, ref2, inp["cumrot"], uv, xyz, ii, inp["time_idx"], inp["K"], conf, lam2, initialize=init)
            assert sa[3] == ntr2 and sa[1] == lam2 and np.abs(sa[0] - ref2).max() / np.abs(ref2).max() < 1e-7, it
        rounds.append(-1)
        # the 20-call schedule chained on the device (one host synchronisation) with two ranks, and with every carried select
        # forced to miss (both ranks fall back alike: the decision is taken on gathered data): the bits of call-by-call stepping
        iters, inits = list(range(20)), [k < 10 for k in range(20)]
        finals = []
        for miss, chained in ((False, False), (False, True), (True, True)):
            e = engine(lo2, hi2)
            if miss:
                e.set_warm_select(2)
            stg = HipStageEngine(e, torch_stream=False)
            stg.attach_rccl(dist, None, FAKE)
            sb = ShardedBA(stg, n, hi2 - lo2, m)
            sb.set_states(st, lam)
            if chained:
                sb.run_schedule(iters, inits)
            else:
                for it, init in zip(iters, inits):
                    sb.step(it, init)
            finals.append(sb.get_states())
            first, misses, lm = stg.stats()
            assert first <= 16 * 1024 and (misses >= 18) == miss and lm > 0, (first, misses, lm)
            sb.close()
        for f in finals[1:]:
            assert np.array_equal(f[0], finals[0][0]) and f[1] == finals[0][1] and f[3] == finals[0][3]
        ref3, lam3 = st.copy(), lam
        for it, init in zip(iters, inits):
            ref3, lam3, _, _ = O.ba_iteration(it, ref3, inp["cumrot"], uv, xyz, ii, inp["time_idx"], inp["K"], conf, lam3, initialize=init)
        assert finals[0][1] == lam3 and np.abs(finals[0][0] - ref3).max() / np.abs(ref3).max() < 1e-6
        if rank == 0:
            np.save(os.path.join(tmp, "rounds.npy"), np.array(rounds))
            single.close()
        a.close()
        c.engine.eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_library_issued_exchanges_with_several_ranks_on_the_test_double_of_rccl(tmp_path, world):
    if not os.path.exists(FAKE) or os.path.getmtime(FAKE) < os.path.getmtime(os.path.join(FAKE_DIR, "fake_rccl.cpp")):
        subprocess.check_call(["make", "-C", FAKE_DIR], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    port = 29300 + (os.getpid() % 200) + 7 * world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rounds = np.load(tmp_path / "rounds.npy")
    assert rounds[:8].max() > 1 and rounds[-1] == -1          # several LM rounds per call; the re-upload leg ran
