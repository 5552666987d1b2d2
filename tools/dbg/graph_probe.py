import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from icem_amd import declared_rssm
for dtype in (None, torch.bfloat16):
    m = declared_rssm(seed=3, dtype=dtype)
    n, h, d = 1024, 12, 6
    dt_ = m.dtype
    o0 = torch.randn(n, 230, device="cuda", dtype=dt_)
    acts = torch.rand(n, h, d, device="cuda", dtype=dt_) * 2 - 1
    out = torch.empty(n, h, device="cuda", dtype=torch.float32)
    def rollout():
        o = o0
        for t in range(h):
            out[:, t] = m.torch_cost(o, acts[:, t]).float()
            o = m.torch_step(o, acts[:, t])
    for _ in range(3): rollout()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): rollout()
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 20
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): rollout()
    torch.cuda.current_stream().wait_stream(s)
    ref = out.clone()
    with torch.cuda.graph(g):
        rollout()
    g.replay(); torch.cuda.synchronize()
    print("graph == eager:", torch.allclose(out, ref))
    t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize(); gr = (time.perf_counter() - t0) / 50
    print(f"{dt_}: eager {eager*1e3:.2f} ms, graph {gr*1e3:.2f} ms per {h}-step rollout of {n}")
