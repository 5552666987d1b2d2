import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner
from oracle import icem_oracle as O
o, d = 378, 17
h = int(sys.argv[1]) if len(sys.argv) > 1 else 6
model = DeviceSyntheticModel.make(o, d, kind=1)
low, high = -0.4 * np.ones(d), 0.4 * np.ones(d)
import os
MODE = int(os.environ.get("WIDE_MODE", "0"))   # icem_set_wide_exact: 0 fp16 planes, 1 exact f32, 2 bf16 planes
SCALE = float(os.environ.get("OBS_SCALE", "0.2"))
for n in (64, 80, 96, 144):
    pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=max(n, 64), opt_iters=1, noise_beta=2.0, dtype="f32", seed=1), low, high)
    pl.set_model(model.kind, model.A, model.B); pl.set_cost(0.1, 2, -1.0, -1, 0.0, 0.0); pl.set_wide_exact(MODE); pl.reset()
    rs = np.random.RandomState(3)
    obs = SCALE * rs.randn(o)
    acts = rs.uniform(-0.4, 0.4, (n, h, d))
    got = pl.rollout_cost(obs, torch.as_tensor(acts, dtype=pl.dt, device="cuda")).cpu().numpy().astype(np.float64)
    om = O.SyntheticModel(model.A, model.B, model.kind)
    want = O.rollout_costs(om, O.CostSpec(ctrl_weight=0.1, lin_idx=2, lin_weight=-1.0, flip_idx=-1, flip_penalty=0.0, flip_thresh=0.0), obs, acts)
    err = np.abs(got - want) / (1 + np.abs(want))
    print(n, "max rel err per 16-row tile:", [float("%.2e" % err[i:i + 16].max()) for i in range(0, n, 16)])
    if n == 80: print("   rows of tile 0:", ["%.1e" % e for e in err[:16]])
