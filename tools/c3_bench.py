import sys, os, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from icem_amd import IcemConfig, IcemPlanner, DeviceSyntheticModel
N, h, d, o = 16384, 30, 17, 24
model = DeviceSyntheticModel.make(o, d, kind=1)
low, high = -0.4 * np.ones(d), 0.4 * np.ones(d)
pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=N, opt_iters=3, noise_beta=2.0, dtype="f32", seed=1), low, high)
pl.set_model(model.kind, model.A, model.B)
pl.set_cost(0.1, 2, -1.0, -1, 0.0, 0.0)   # HumanoidStandup: -obs[2] + 0.1 |a|^2
pl.reset()
pl.obs0.copy_(torch.as_tensor(0.1 * np.random.RandomState(0).randn(o), dtype=pl.dt))
for _ in range(5): pl.plan_step_resident()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): pl.plan_step_resident()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
pl.profile_enable(True)
for _ in range(10): pl.plan_step_resident()
torch.cuda.synchronize()
prof = pl.profile_read()
ts = sum(pl.population_sizes) * h
print(f"C3-shaped (N={N}, h={h}, d={d}, o={o}, beta=2, tanh model, 3 iters): {dt*1e6:.1f} us/MPC step, {ts/dt/1e9:.2f} G traj-steps/s;",
      ", ".join(f"{k} {1e3*v[0]/v[1]:.1f} us x{v[1]//10}" for k, v in prof.items()))
