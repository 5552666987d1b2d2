"""Phase stamps (wall_clock64, 100 MHz) of tile 0 of the split learned-dynamics launch (icem_rssm_split.hip): the
recurrence workgroup's step 5 phase by phase, and where the reward-head workgroup is at the same moments."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icem_amd import DeviceRSSMModel  # noqa: E402
from icem_amd import _lib as L  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
h = 12
m = DeviceRSSMModel(seed=3)
obs = 0.3 * np.random.RandomState(1).randn(230)
acts = torch.rand(n, h, 6, device="cuda") * 2 - 1
dbg = torch.zeros(16, dtype=torch.int64, device="cuda")
L.check(m.lib.icem_debug_stamps(None, C.c_void_p(dbg.data_ptr())))
for _ in range(5):
    m.rollout_cost(obs, acts)
torch.cuda.synchronize()
R = 20
acc = np.zeros(16)
for _ in range(R):
    m.rollout_cost(obs, acts)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().astype(np.float64)
    acc += (d - d[0]) / 100.0
acc /= R
names = ["entry", "init done", "step5 top", "step5 after x", "step5 after GRU", "step5 after p", "step5 after z'", "recurrence done",
         "reward: entry", "reward: weights resident", "reward: state 5 seen", "reward: state 5 scored", "reward: last state seen", "reward: done"]
for k, nm in enumerate(names):
    print(f"{nm:28s} {acc[k]:8.2f} us")
print("step 5 phases (us): x %.2f  GRU %.2f  p %.2f  z' %.2f  | step %.2f" % (acc[3] - acc[2], acc[4] - acc[3], acc[5] - acc[4], acc[6] - acc[5], acc[6] - acc[2]))
L.check(m.lib.icem_debug_stamps(None, None))
