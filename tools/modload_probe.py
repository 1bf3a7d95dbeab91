import sys, time
sys.path.insert(0, ".")
import numpy as np
from quartetscores_amd import engine
ctx = engine.Context(64, 32)
ctx.table_alloc(); ctx.sync()
q = np.array([[0, 1, 2, 3]], dtype=np.uint16)
for i in range(3):
    t0 = time.perf_counter(); ctx.lookup(q); t1 = time.perf_counter()
    print(f"qs_lookup call {i}: {(t1 - t0) * 1e3:.3f} ms")
