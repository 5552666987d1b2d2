"""Door / Relocate / FetchPickAndPlace shapes (h = 30; the reference's settings/{door,relocate,fpp}) on the TileHN kernel
(k_rollout_hn.hip) against the exact-f32 GEMM kernel (icem_set_tile_arith 0) and the float64 oracle: cost errors of a
stand-alone rollout, us per MPC step in both, HalfCheetah's beside them.  usage: python tools/dbg/hn_check.py [N ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env
from icem_amd import envs as E
from oracle import icem_oracle as O

ENVS = {"door": (E.door_env, O.CostSpec.door, 2.5), "relocate": (E.relocate_env, O.CostSpec.relocate, 3.5),
        "fpp": (E.fetch_pick_and_place_env, O.CostSpec.fetch_pick_and_place, 3.0)}


def planner(env, model, N, beta, arith, iters=5):
    pl = IcemPlanner(IcemConfig(horizon=30, act_dim=env.action_space.shape[0], num_traj=N, opt_iters=iters, dtype="f32", seed=1, noise_beta=beta),
                     env.action_space.low, env.action_space.high)
    pl.set_model(model.kind, model.A, model.B)
    pl.set_cost_spec(env.cost_spec)
    pl.set_tile_arith(arith)
    pl.reset()
    pl.obs0.copy_(torch.as_tensor(0.1 * np.random.RandomState(0).randn(env.obs_dim), dtype=pl.dt))
    return pl


def timed(pl, steps=100):
    for _ in range(10):
        pl.plan_step_resident()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        pl.plan_step_resident()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


def main():
    Ns = [int(x) for x in sys.argv[1:]] or [4096]
    for name, (mk, spec_fn, beta) in ENVS.items():
        env, spec = mk(), spec_fn()
        o, d = env.obs_dim, env.action_space.shape[0]
        for kind in (0, 1):
            model = DeviceSyntheticModel.make(o, d, kind=kind)
            om = O.SyntheticModel(model.A, model.B, model.kind)
            rs = np.random.RandomState(3)
            obs0 = 0.2 * rs.randn(o)
            acts = rs.uniform(-1, 1, (531, 30, d)) * env.action_space.high
            want = O.rollout_costs(om, spec, obs0, acts).astype(np.float64)
            line = f"{name:9s} d={d:2d} o={o:2d} kind={kind}:"
            for arith in (-1, 0):
                pl = planner(env, model, Ns[0], beta, arith)
                got = pl.rollout_cost(obs0, torch.as_tensor(acts, dtype=pl.dt, device=pl.device)).cpu().numpy().astype(np.float64)
                err = np.abs(got - want) / (1 + np.abs(want))
                line += f"  arith {pl.tile_arith}: median rel err {np.median(err):.1e}, > 1e-4: {(err > 1e-4).mean():.3f}"
            print(line, flush=True)
        model = DeviceSyntheticModel.make(o, d, kind=1)
        for N in Ns:
            pp = planner(env, model, N, beta, -1)
            for _ in range(5):
                pp.plan_step_resident()
            torch.cuda.synchronize()
            pp.profile_enable(True)
            for _ in range(10):
                pp.plan_step_resident()
            torch.cuda.synchronize()
            print(f"{name:9s} N={N}: us per launch (launches per step):", {k: (round(1e3 * v[0] / v[1], 1), v[1] // 10) for k, v in pp.profile_read().items()}, flush=True)
            a, b = timed(planner(env, model, N, beta, -1)), timed(planner(env, model, N, beta, 0))
            print(f"{name:9s} N={N}: TileHN {a:8.1f} us per MPC step   exact GEMM kernel {b:8.1f}   ({b / a:.2f}x)", flush=True)
    env = halfcheetah_env(17)
    model = DeviceSyntheticModel.make(17, 6, kind=1)
    for N in Ns:
        print(f"halfcheetah N={N}: {timed(planner(env, model, N, 0.25, -1)):8.1f} us per MPC step", flush=True)


if __name__ == "__main__":
    main()
