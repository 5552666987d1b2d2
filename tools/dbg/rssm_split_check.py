"""The split launch (recurrence + reward-head workgroups, icem_rssm_split.hip) against the fused kernel: costs must be
bit-identical.  Run once with ICEM_RSSM_SPLIT=0 to write the fused kernel's costs, once without to compare.
usage: rssm_split_check.py write|check <file.npz>"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icem_amd import DeviceRSSMModel  # noqa: E402
from icem_amd import _lib as _LENV  # noqa: E402
_LENV.follow_environment()   # this tool flips ICEM_<NAME> variables: mapped onto icem_set_option per planner (the library reads no environment)

CASES = [(1, 12, 0), (15, 1, 2), (16, 2, 0), (17, 12, 1), (1007, 12, 0), (1024, 12, 1), (1024, 12, 2), (2048, 12, 0), (2033, 30, 0),
         (640, 5, 1), (4096, 12, 0), (3001, 12, 1), (4097, 12, 0), (8192 + 21, 12, 1), (20000, 12, 0), (65536, 12, 0), (40001, 3, 2)]
m = DeviceRSSMModel(seed=3)
out = {}
for rep in range(3):   # (repeats: the flags must come back to 0 behind every launch)
    for (n, h, mode) in CASES:
        rs = np.random.RandomState(n + 7 * h + mode + 100 * rep)
        acts = torch.as_tensor(rs.uniform(-1, 1, (n, h, 6)), dtype=torch.float32, device="cuda")
        obs = 0.3 * rs.randn(230)
        out[f"{n}_{h}_{mode}_{rep}"] = m.rollout_cost(obs, acts, mode).cpu().numpy()
if sys.argv[1] == "write":
    np.savez(sys.argv[2], **out)
    print("wrote", len(out), "cases")
else:
    ref = np.load(sys.argv[2])
    bad = [k for k in out if not np.array_equal(out[k], ref[k], equal_nan=True)]
    for k in bad:
        d = np.flatnonzero(out[k] != ref[k])
        print("MISMATCH", k, np.abs(out[k] - ref[k]).max(), np.isnan(out[k]).sum(), "rows", len(d), d[:12], "tiles", sorted(set(d // 16))[:12])
    print(f"split vs fused: {len(out)} cases, {len(bad)} differ")
    sys.exit(1 if bad else 0)
