"""Observation-sharded mode with the exchanges issued by the LIBRARY (vba_sh_comm_init / vba_sh_call: RCCL all-gathers on
the handle's stream, RCCL resolved at run time from the copy torch has loaded).  A one-GPU box can give RCCL one rank only
(one device per rank), so this runs the whole native path -- unique id over a gloo control group, communicator, padded
slots, three all-gathers per LM round -- at world size 1, against the caller-dispatched protocol (same kernels, torch's
all_gather_into_tensor) bit for bit, the unsharded engine and the oracle.  What N ranks add is RCCL's own business: the
buffers, counts and order of the three ncclAllGather calls are those of all_gather_into_tensor in vinsat_amd/dist.py, which
the gloo tests (world 2, 3) and the two-process GPU test cover."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from conftest import golden_inputs, load_golden
    from oracle import ba_oracle as O
    from vinsat_amd.dist import HipStageEngine, ShardedBA, loaded_rccl_path
    from vinsat_amd.engine import BAEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        assert "librccl" in loaded_rccl_path()
        g = load_golden("c2")
        inp = golden_inputs(g)
        n, m = inp["K"].shape[0], inp["xyz"].shape[0]
        conf = inp["conf"].copy()
        conf[:] = 3.0                        # the LM loop rejects trials: several rounds (and all-gathers) per call

        def make(native):
            eng = BAEngine(n, m)
            eng.upload_observations(inp["xyz"], inp["uv"], conf, inp["ii"], n)
            eng.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
            stage = HipStageEngine(eng, torch_stream=not native)
            if native:
                stage.attach_rccl(dist)
            else:
                nccl = dist.new_group(backend="nccl", device_id=torch.device("cuda", 0))
                return ShardedBA(stage, n, m, m, group=nccl)
            return ShardedBA(stage, n, m, m)

        a, b = make(True), make(False)
        assert a.engine.native and not b.engine.native
        single = BAEngine(n, m)
        single.upload_observations(inp["xyz"], inp["uv"], conf, inp["ii"], n)
        single.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
        st, lam = g["states0"][0], 1e-4
        a.set_states(st, lam)
        b.set_states(st, lam)
        ref, lam_ref, sg, lam_g = st.copy(), lam, st.copy(), lam
        rounds = []
        for it, init in [(0, True), (1, True), (2, True), (5, True), (10, False), (11, False), (12, False), (13, False)]:
            na, nb = a.step(it, init), b.step(it, init)
            sa, sb = a.get_states(), b.get_states()
            assert na == nb and np.array_equal(sa[0], sb[0]) and sa[1] == sb[1] and sa[3] == sb[3], it     # same kernels, another dispatcher
            ref, lam_ref, _, ntr_ref = O.ba_iteration(it, ref, inp["cumrot"], inp["uv"], inp["xyz"], inp["ii"], inp["time_idx"], inp["K"], conf,
                                                      lam_ref, initialize=init)
            assert sa[3] == ntr_ref and sa[1] == lam_ref and np.abs(sa[0] - ref).max() / np.abs(ref).max() < 1e-7
            sg, lam_g, _, ntr_g, _ = single.iterate(it, init, lam_g, sg)
            assert ntr_g == sa[3] and lam_g == sa[1] and np.abs(sa[0] - sg).max() / np.abs(sg).max() < 1e-9
            rounds.append(na)
        np.save(os.path.join(tmp, "rounds.npy"), np.array(rounds))
        # a second communicator on a handle is refused; a call without one as well
        with pytest.raises(Exception):
            a.engine.attach_rccl(dist)
        with pytest.raises(Exception):
            b.engine.call(0, True, m)
        a.close(); b.close(); single.close()
    finally:
        dist.destroy_process_group()


def _worker_protocols(rank, world, port, tmp):
    """Carried-keys protocol (default) against the round-3 protocol and the unsharded engine: call by call, the 20-call schedule
    chained on the device (vba_sh_run_schedule), every carried select forced to miss (fallback: exact select over all keys)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from conftest import golden_inputs, load_golden
    from vinsat_amd.dist import HipStageEngine, ShardedBA
    from vinsat_amd.engine import BAEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        g = load_golden("c2")
        inp = golden_inputs(g)
        n, m = inp["K"].shape[0], inp["xyz"].shape[0]
        iters, inits = list(range(20)), [k < 10 for k in range(20)]
        out = {}
        for confname, conf in (("golden", inp["conf"]), ("rejecting", np.full(m, 3.0))):
            def make(proto, force_miss=False, slack=0):
                eng = BAEngine(n, m + slack)        # (slack: a handle NOT sized for ceil(m_total / ranks) rows takes the round-3 protocol)
                eng.upload_observations(inp["xyz"], inp["uv"], conf, inp["ii"], n)
                eng.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
                if force_miss:
                    eng.set_warm_select(2)
                stage = HipStageEngine(eng, torch_stream=False)
                stage.attach_rccl(dist)
                stage.set_protocol(proto)
                return ShardedBA(stage, n, m, m)

            res = {}
            for name, proto, miss, chained in (("carried", 1, False, False), ("carried-chained", 1, False, True), ("round3", 0, False, False),
                                               ("carried-missing", 1, True, True), ("oversized-handle", 1, False, True)):
                sb = make(proto, miss, 300 if name == "oversized-handle" else 0)
                sb.set_states(g["states0"][0], 1e-4)
                if chained:
                    sb.run_schedule(iters[:7], inits[:7])          # two schedules: the second starts from carried keys
                    sb.run_schedule(iters[7:], inits[7:])
                else:
                    for it, init in zip(iters, inits):
                        sb.step(it, init)
                res[name] = sb.get_states()
                first, misses, lm = sb.engine.stats()
                if name == "oversized-handle":
                    assert first == 16 * m and misses == 0      # (its exchange buffers would not match other ranks': round-3 protocol)
                elif proto == 1:
                    assert first <= 16 * 1024, first                # the first exchange of a call: histogram + block sums, not 16 B per row
                    assert (misses >= 18) == miss, (name, misses)   # forced misses: every carried call fell back (the first has no carried keys)
                    assert (lm > 0) == (confname == "rejecting"), (name, lm)
                else:
                    assert first == 16 * m
                sb.close()
            # unsharded calls in between on the same handle: the gathered exchange of the last sharded trial must not be reused
            sb = make(1)
            sb.set_states(g["states0"][0], 1e-4)
            sb.run_schedule(iters[:6], inits[:6])
            sb.engine.eng.step(iters[6], inits[6])
            sb.engine.eng.step(iters[7], inits[7])
            sb.run_schedule(iters[8:], inits[8:])
            res["mixed-with-unsharded-calls"] = sb.get_states()
            sb.close()
            single = BAEngine(n, m)
            single.upload_observations(inp["xyz"], inp["uv"], conf, inp["ii"], n)
            single.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
            single.set_states(g["states0"][0], 1e-4)
            single.run_schedule(iters, inits)
            ref = single.get_states()
            single.close()
            mixed = res.pop("mixed-with-unsharded-calls")
            assert mixed[1] == ref[1] and np.abs(mixed[0] - ref[0]).max() / np.abs(ref[0]).max() < 1e-9, confname
            base = res["carried"]
            for name, r in res.items():
                assert np.array_equal(r[0], base[0]) and r[1] == base[1] and r[3] == base[3], (confname, name)     # one set of bits
            assert base[1] == ref[1] and base[3] == ref[3] and np.abs(base[0] - ref[0]).max() / np.abs(ref[0]).max() < 1e-9, confname
            if confname == "golden":
                assert np.abs(base[0] - g["states_out_19"][0]).max() / np.abs(g["states_out_19"][0]).max() < 1e-6
            out[confname] = base[3]
        np.save(os.path.join(tmp, "ntr.npy"), np.array([out["golden"], out["rejecting"]]))
    finally:
        dist.destroy_process_group()


def test_carried_keys_protocol_chained_stepped_and_falling_back_gives_one_set_of_bits(tmp_path):
    port = 29100 + (os.getpid() % 200)
    mp.spawn(_worker_protocols, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    assert np.load(tmp_path / "ntr.npy").shape == (2,)


def test_library_issued_rccl_exchanges_match_the_caller_dispatched_protocol(tmp_path):
    port = 29500 + (os.getpid() % 200)
    mp.spawn(_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    rounds = np.load(tmp_path / "rounds.npy")
    assert rounds.shape == (8,) and rounds.max() > 1
