"""Run-to-run determinism of the split-bf16 wide rollout: the same launch many times, outputs compared bitwise."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner
from icem_amd import envs as E
o, d, h = 376, 17, 12
env = E.humanoid_env(healthy_z_range=(-0.05, 2.0))
model = DeviceSyntheticModel.make(o, d, kind=1)
for n in [int(x) for x in sys.argv[1:]] or [640, 409, 257, 100, 1040]:
    pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=max(n, 64), opt_iters=1, dtype="f32", seed=7), env.action_space.low, env.action_space.high)
    pl.set_model(model.kind, model.A, model.B); pl.set_cost_spec(env.cost_spec); pl.reset()
    rs = np.random.RandomState(3)
    obs0 = 0.2 * rs.randn(o)
    acts = torch.as_tensor(rs.uniform(-1, 1, (n, h, d)) * env.action_space.high, dtype=pl.dt, device=pl.device)
    ref = pl.rollout_cost(obs0, acts).clone()
    bad = 0; rows = set()
    junk = torch.empty(64 << 20, device="cuda")
    for i in range(300):
        if i % 7 == 0: junk.normal_()          # other kernels in between: LDS and registers are left dirty
        got = pl.rollout_cost(obs0, acts)
        ne = (got != ref) & ~(torch.isnan(got) & torch.isnan(ref))
        if ne.any():
            bad += 1; rows.update(ne.nonzero().flatten().tolist()[:8])
    print("n=%5d: %d of 300 launches differ from the first; rows %s" % (n, bad, sorted(rows)[:16]))
