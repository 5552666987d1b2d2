import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from icem_amd import DeviceRSSMModel
m = DeviceRSSMModel(seed=3)
n, h, d = 64, 12, 6
rs = np.random.RandomState(4)
acts = rs.uniform(-1, 1, (n, h, d)); obs = 0.3 * rs.randn(230)
A = torch.as_tensor(acts, dtype=torch.float32, device="cuda")
per = np.stack([m.rollout_cost(obs, A[:, :t + 1].contiguous(), 2).cpu().numpy() for t in range(h)], 1)   # c_t via 'final' of a (t+1)-step rollout
tot = m.rollout_cost(obs, A, 0).cpu().numpy()
fin = m.rollout_cost(obs, A, 2).cpu().numpy()
print("sum vs sum of finals:", np.abs(tot - per.sum(1)).max(), " final vs last:", np.abs(fin - per[:, -1]).max())
net = m.reference.double().cpu()
o = torch.as_tensor(obs).expand(n, -1).double(); a = torch.as_tensor(acts).double()
ref = []
with torch.no_grad():
    for t in range(h):
        ref.append(-net.reward(o).numpy()); o = net(o, a[:, t])
ref = np.stack(ref, 1)
print("per-step max err vs exact net:", np.abs(per - ref).max(0).round(4))
