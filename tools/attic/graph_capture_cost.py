"""Cost of capturing a chained schedule as a graph: run_schedule of the 20-call schedule four times in a row, graph off and on (us)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vinsat_amd import od_pipe, synth
from vinsat_amd.engine import BAEngine
win = od_pipe.prepare_window(*synth.make_sequence("C3")); st0 = od_pipe.initial_guess(win)
n, m = win.time_idx.size, win.ii.size
for graph in (False, True):
    e = BAEngine(n, m); e.set_schedule_graph(graph)
    e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n); e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
    it, ini = list(range(20)), [k < 10 for k in range(20)]
    ts = []
    for r in range(4):
        e.set_states(st0, 1e-4); t0 = time.perf_counter(); e.run_schedule(it, ini); ts.append(1e6 * (time.perf_counter() - t0))
    print("graph" if graph else "stream", [round(t) for t in ts], e.schedule_graph_stats(), flush=True)
    e.close()
