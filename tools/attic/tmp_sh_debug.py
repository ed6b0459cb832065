import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.distributed as dist
from conftest import golden_inputs, load_golden
from vinsat_amd.dist import HipStageEngine, ShardedBA
from vinsat_amd.engine import BAEngine
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29911")
dist.init_process_group("gloo", rank=0, world_size=1)
torch.cuda.set_device(0)
g = load_golden("c2"); inp = golden_inputs(g)
n, m = inp["K"].shape[0], inp["xyz"].shape[0]
conf = inp["conf"].copy()
if len(sys.argv) > 1: conf[:] = float(sys.argv[1])
def make(proto):
    eng = BAEngine(n, m)
    eng.upload_observations(inp["xyz"], inp["uv"], conf, inp["ii"], n)
    eng.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
    st = HipStageEngine(eng, torch_stream=False)
    st.attach_rccl(dist)
    st.set_protocol(proto)
    return ShardedBA(st, n, m, m)
a, b = make(True), make(False)
st = g["states0"][0]
a.set_states(st, 1e-4); b.set_states(st, 1e-4)
for it, init in [(0, True), (1, True), (2, True), (5, True), (10, False), (11, False), (12, False), (13, False)]:
    na, nb = a.step(it, init), b.step(it, init)
    sa, sb = a.get_states(), b.get_states()
    print(it, init, "rounds", na, nb, "ntr", sa[3], sb[3], "lam", sa[1], sb[1], "maxdiff", np.abs(sa[0] - sb[0]).max(), "stats", a.engine.stats(), flush=True)
