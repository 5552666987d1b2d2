"""Times icem_trajectory_cost (trajectory_cost_fn over rollouts an external model left in HBM) at the reference's
environment shapes and reports achieved HBM bandwidth against the bytes each cost actually has to read:
the full observation rows when the health term is on (all_finite over the row), otherwise only the actions and
the handful of observation entries the cost reads."""
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from icem_amd import IcemConfig, IcemPlanner  # noqa: E402
from icem_amd import envs as E  # noqa: E402


def run(name, env, n, h, reps=20):
    o, d = env.obs_dim, env.action_space.shape[0]
    pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=n, opt_iters=1, dtype="f32"), env.action_space.low,
                     env.action_space.high)
    pl.set_cost_spec(env.cost_spec)
    obs = torch.randn((n, h, o), device=pl.device)
    nxt = torch.randn((n, h, o), device=pl.device) if env.cost_spec.diff_idx >= 0 else None
    act = torch.rand((n, h, d), device=pl.device) * 2 - 1
    for _ in range(3):
        pl.trajectory_cost(obs, act, nxt)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        pl.trajectory_cost(obs, act, nxt)
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 1e3 / reps
    sweep = env.cost_spec.health_idx >= 0
    by = 4.0 * n * h * (d + (o if sweep else 0)) + 4.0 * n
    print(f"{name:22s} n={n} h={h} o={o} d={d}: {us:8.1f} us  {by / us / 1e6:7.2f} TB/s of {by / 1e6:8.1f} MB "
          f"({'row sweep' if sweep else 'entries only'})")


if __name__ == "__main__":
    run("Humanoid", E.humanoid_env(376), 16384, 30)
    run("Ant", E.ant_env(), 16384, 30)
    run("HumanoidStandup o=378", E.humanoid_standup_env(378), 16384, 30)
    run("Hopper", E.hopper_env(), 65536, 30)
    run("FetchPickAndPlace", E.fetch_pick_and_place_env(), 65536, 30)
