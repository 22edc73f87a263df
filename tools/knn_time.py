import numpy as np, torch, time, sys
sys.path.insert(0,'.')
from simple_knn._C import distCUDA2
rng = np.random.default_rng(5)
c = rng.standard_normal((2000, 3)) * 5
pts = (c[rng.integers(0, 2000, 1_000_000)] + 0.3 * rng.standard_normal((1_000_000, 3))).astype(np.float32)
x = torch.from_numpy(pts).cuda()
for P in (100_000, 1_000_000):
    y = distCUDA2(x[:P]); torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(5): y = distCUDA2(x[:P])
    torch.cuda.synchronize(); print(P, "distCUDA2 ms", (time.perf_counter()-t)/5*1e3)
