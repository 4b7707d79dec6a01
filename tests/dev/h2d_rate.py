"""Diagnostic (GPU box): host-to-device and device-to-host copy rates, pageable against pinned, for the sizes of the host entry's
pieces (8 MB cost blocks up, 3.3 MB byte tables down)."""
import time
import numpy as np
import torch
dev = torch.device("cuda", 0)
for mb in (8, 33.5):
    n = int(mb * 1e6 / 8)
    a = np.random.rand(n)
    d = torch.empty(n, dtype=torch.float64, device=dev)
    pin = torch.empty(n, dtype=torch.float64).pin_memory()
    pin.copy_(torch.from_numpy(a))
    for name, src in (("pageable", torch.from_numpy(a)), ("pinned", pin)):
        for _ in range(3):
            d.copy_(src)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            d.copy_(src)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"H2D {mb} MB {name}: {dt*1e3:.3f} ms = {mb/1e3/dt:.1f} GB/s")
    t0 = time.perf_counter()
    for _ in range(10):
        pin.copy_(torch.from_numpy(a))
    dt = (time.perf_counter() - t0) / 10
    print(f"host memcpy {mb} MB pageable -> pinned, one thread: {dt*1e3:.3f} ms = {mb/1e3/dt:.1f} GB/s")
