"""Host->device upload rate of the data matrix (config B, 800 MB) as the drop-in functions see it with NumPy input."""
import time, numpy as np, torch
X = np.random.rand(100000, 2000).astype(np.float32)
torch.cuda.init(); torch.cuda.synchronize()
for pinned in (False, True):
    t = torch.from_numpy(X)
    if pinned:
        t = t.pin_memory()
    for _ in range(2):
        d = t.to("cuda"); torch.cuda.synchronize()
    t0 = time.time(); d = t.to("cuda"); torch.cuda.synchronize(); dt = time.time() - t0
    print(f"pinned={pinned}: {dt*1e3:.1f} ms for {X.nbytes/1e6:.0f} MB -> {X.nbytes/dt/1e9:.1f} GB/s")
