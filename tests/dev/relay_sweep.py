"""Sweep of the relay's two knobs on one config: pieces x first cut (KBEST_RELAY x KBEST_RELAY_FIRST / 1024 of k).
python3 tests/dev/relay_sweep.py c4|c3 [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl

dev = torch.device("cuda", 0)
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
Bq = int(sys.argv[2]) if len(sys.argv) > 2 else None


def engine(**env):
    for k_, v in env.items():
        os.environ[k_] = str(v)
    e = pk.KBestEngine(0)
    for k_ in env:
        del os.environ[k_]
    return e


costs, N, M, k = wl.dense_config(cfg, B=Bq)
B = costs.shape[0]
d_cost = torch.from_numpy(costs).to(dev)
d_r = torch.empty((B, k, N), dtype=torch.int32, device=dev); d_c = torch.empty((B, k, N), dtype=torch.int32, device=dev)
d_g = torch.empty((B, k), dtype=torch.float64, device=dev); d_n = torch.empty(B, dtype=torch.int32, device=dev)
st = torch.cuda.Stream()
pieces = [int(x) for x in os.environ.get("SWEEP_P", "2,3,4,5,6,8").split(",")]
firsts = [int(x) for x in os.environ.get("SWEEP_F", "256,384,512,640").split(",")]
step = os.environ.get("SWEEP_STEP")  # with it: the columns are steps (later cuts, / 1024 of k apart) at the ONE first cut SWEEP_F names
names, engs = ["plain"], [engine(KBEST_RELAY=0)]
if step:
    F0 = firsts[0]
    firsts = [int(x) for x in step.split(",")]
for P in pieces:
    for F in firsts:
        names.append(f"P{P} F{F}")
        engs.append(engine(KBEST_RELAY=P, KBEST_RELAY_FIRST=F0, KBEST_RELAY_STEP=F) if step else engine(KBEST_RELAY=P, KBEST_RELAY_FIRST=F))
res = {n: [] for n in names}
for rnd in range(3):
    for n, e in zip(names, engs):
        ts = []
        for it in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(st):
                a.record(); e.kbest_dev(d_cost, B, N, M, k, d_r, d_c, d_g, d_n, stream=st.cuda_stream); b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        res[n].append(min(ts[1:]))
print(f"{cfg} B={B}: plain {np.median(res['plain']):.3f}")
for P in pieces:
    print(f"  {P} pieces, " + (f"first cut {F0}, steps " if step else "first cut ") + "  ".join(f"{F}/1024: {np.median(res[f'P{P} F{F}']):.3f}" for F in firsts), flush=True)
