"""HBM read bandwidth a plain streaming kernel reaches on buffers of several sizes (torch reductions / copies)."""
import time
import torch
for gb in (2.1, 8.0, 34.0):
    n = int(gb * 1e9 / 4)
    x = torch.ones(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for name, f in (("sum(int32)", lambda: x.sum()), ("max(int32)", lambda: x.max()), ("copy 1/2", lambda: x[: n // 2].copy_(x[n // 2:]))):
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        moved = gb if not name.startswith("copy") else gb   # copy: reads half, writes half
        print(f"{gb:5.1f} GB  {name:12s} {best * 1e3:8.3f} ms  {moved / best / 1e3:6.2f} TB/s", flush=True)
    del x
    torch.cuda.empty_cache()
