import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icem_amd import DeviceRSSMModel
m = DeviceRSSMModel(seed=3)
obs = 0.3 * np.random.RandomState(1).randn(230)
rs = np.random.RandomState(5)
bad = 0
for n in (1000, 3000, 9000):
    a1 = torch.as_tensor(rs.uniform(-1, 1, (n, 12, 6)), dtype=torch.float32, device="cuda")
    a2 = torch.as_tensor(rs.uniform(-1, 1, (n, 12, 6)), dtype=torch.float32, device="cuda")
    r1 = m.rollout_cost(obs, a1).clone(); r2 = m.rollout_cost(obs, a2).clone()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(20):
        with torch.cuda.stream(s1):
            o1 = m.rollout_cost(obs, a1)
        with torch.cuda.stream(s2):
            o2 = m.rollout_cost(obs, a2)
        torch.cuda.synchronize()
        bad += int(not torch.equal(o1, r1)) + int(not torch.equal(o2, r2))
print("two streams concurrently:", bad, "mismatching launches")
