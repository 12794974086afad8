"""Experiment: why does bench.py's real-sample leg measure 194 k frames/s at depth 4 where tools/bench_samples.py measures 107 k?
-> which frames share a launch (usage: exp_samples_order.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fixtures as fx
import squad_mortar_helper_amd as smh

vision = smh.HipVision.init(0)
batch = 128


def load(only_1440p_checked_first):
    out = []
    for stem in fx.OPEN_STEMS:
        f = fx.load_fixture(stem)[0]
        if f.shape[:2] == (1440, 2560):
            out.append(f)
    return out


def rate(d, dep, steps=200, **kw):
    pipe = smh.Pipeline(vision, 2560, 1440, batch, depth=dep, **kw)
    for _ in range(2 * dep):
        pipe.submit(d.data_ptr(), batch, stages=3, max_gap=15)
    pipe.wait(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        s = pipe.submit(d.data_ptr(), batch, stages=3, max_gap=15)
    pipe.wait(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    recs = smh.results_to_dicts(pipe.slots[s].read_results(0, batch))
    pipe.close()
    return round(batch * steps / dt), sum(r["rounds"] for r in recs), sum(r["map_open"] for r in recs)


fr = load(True)
k = len(fr)
print("frames:", k, [f.shape for f in fr[:3]])
d = torch.from_numpy(np.stack([fr[i % k] for i in range(batch)])).cuda()
print("d4 (rate, total rounds, open maps):", rate(d, 4), " explicit batch:", rate(d, 4, search="batch"), " d12:", rate(d, 12))
allf = [fx.load_fixture(s)[0] for s in fx.OPEN_STEMS]
print("all OPEN_STEMS shapes:", sorted(set(f.shape for f in allf)), len(allf))
