"""What the term list costs a TileHN launch: the Door / Relocate / FetchPickAndPlace shapes at N = 4096 with the env's cost terms
and with icem_cost_spec's form alone (no terms), per wave arrangement.  usage (GPU box): python tools/dbg/hn_terms_time.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner
from icem_amd import envs as E
from icem_amd import _lib as _LENV  # noqa: E402
_LENV.follow_environment()   # this tool flips ICEM_<NAME> variables: mapped onto icem_set_option per planner (the library reads no environment)
N = 4096
for name, mk in (("door", E.door_env), ("relocate", E.relocate_env), ("fpp", E.fetch_pick_and_place_env)):
    env = mk()
    o, d = env.obs_dim, env.action_space.shape[0]
    model = DeviceSyntheticModel.make(o, d, kind=1)
    for terms in (True, False):
        for label, ev in (("split", {}), ("pair", {"ICEM_HN_SPLIT": "0"}), ("single", {"ICEM_HN_PAIR": "0"})):
            for k in ("ICEM_HN_SPLIT", "ICEM_HN_PAIR"):
                os.environ.pop(k, None)
            os.environ.update(ev)
            pl = IcemPlanner(IcemConfig(horizon=30, act_dim=d, num_traj=N, opt_iters=2, dtype="f32", seed=1), env.action_space.low, env.action_space.high)
            pl.set_model(model.kind, model.A, model.B)
            if terms:
                pl.set_cost_spec(env.cost_spec)
            else:
                pl.set_cost(0.1, 0, -1.0, 1, 10.0, 0.3)
                pl.set_tile_arith(1)
            act = (torch.rand((N, 30, d), dtype=pl.dt, device=pl.device) * 2 - 1) * float(env.action_space.high[0])
            obs = 0.1 * np.random.RandomState(0).randn(o)
            for _ in range(3):
                pl.rollout_cost(obs, act)
            torch.cuda.synchronize()
            pl.profile_enable(True)
            for _ in range(20):
                pl.rollout_cost(obs, act)
            torch.cuda.synchronize()
            prof = pl.profile_read()
            pl.profile_enable(False)
            us = {k: round(1e3 * v[0] / v[1], 1) for k, v in prof.items()}
            print(f"{name:9s} terms={terms!s:5s} {label:6s} tile_arith={pl.tile_arith}: {us}")
