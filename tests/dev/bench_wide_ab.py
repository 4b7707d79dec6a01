"""A/B of the general-size kernel's launch knobs (KBEST_WIDE_NW / KBEST_WIDE_TILE / KBEST_WIDE_SPEC).  Development aid."""
import os, sys, subprocess
here = os.path.abspath(__file__)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(here))))
    import torch
    dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
    import probabilisticsemslam_amd as pk
    rng = np.random.default_rng(0)
    N, M, k, B = (int(x) for x in sys.argv[2:6])
    eng = pk.KBestEngine(0)
    scale = float(os.environ.get("SCALE", "50"))
    if os.environ.get("SPLITMIX"):
        from probabilisticsemslam_amd import workloads as wl
        costs = torch.from_numpy(wl.dense_batch(B, N, M, 0x5EED0000 + 1000 * N + k)).to(dev)
    else:
        costs = torch.from_numpy(rng.random((B, N * M)) * scale).to(dev)
    r4c = torch.empty((B, k, M), dtype=torch.int32, device=dev); c4r = torch.empty((B, k, N), dtype=torch.int32, device=dev)
    g = torch.empty((B, k), dtype=torch.float64, device=dev); nf = torch.empty(B, dtype=torch.int32, device=dev)
    ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts); s = ts.cuda_stream
    eng.kbest_dev(costs, B, N, M, k, r4c, c4r, g, nf, stream=s); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): eng.kbest_dev(costs, B, N, M, k, r4c, c4r, g, nf, stream=s)
    e1.record(); torch.cuda.synchronize()
    print(f"{e0.elapsed_time(e1)/3:.2f} ms  gsum {float(g.sum()):.9e}")
    sys.exit(0)
shapes = [(128, 128, 200, 256), (128, 128, 200, 512), (64, 64, 200, 256), (256, 256, 200, 256)]
variants = [{"KBEST_WIDE_SPEC": str(x)} for x in (2, 3, 4, 6, 8)]
for sh in shapes:
    for v in variants:
        env = dict(os.environ); env.update(v)
        if sh[0] == 64: env["KBEST_FORCE_WIDE"] = "1"
        r = subprocess.run([sys.executable, here, "child"] + [str(x) for x in sh], env=env, capture_output=True, text=True, timeout=300)
        print(sh, v, r.stdout.strip() or r.stderr.strip()[-300:], flush=True)
