#!/usr/bin/env python3
"""bench.py -- iCEM inner planning loop on MI355X: traj-steps/s of whole MPC steps.

  python bench.py --gpus N --steps K --warmup W [--workload c2|c3|c4|c5]

A "step" is one MPC step = all CEM iterations (sample -> rollout -> cost -> top-k -> refit) of one
`get_action`, on synthetic HalfCheetah-shaped input already resident in HBM.  Workload c2 (default;
BASELINE.json's metric is quoted on it): N=4096, h=30, d=6, o=17, beta=0.25, 5 iterations with
population decay (4096, 3276, 2620, 2096, 1676), K=10, f32.  Workload c4: N=65536, same otherwise.  Workload c3:
HumanoidStandup action shapes (N=16384, d=17, beta=2.0, 3 iterations) on a 24-dim tanh latent model.  Workload c5
(single GPU): the learned-dynamics configuration N=1024, h=12 -- controller-driven steps with the declared RSSM's rollout
fused on the bf16 matrix cores; its `roofline` is bound "mfma" and its CPU baseline oracle/rssm_oracle.py.
For --gpus G > 1 there is one rank per GPU: either the driver launches them (torch.distributed.run sets RANK /
WORLD_SIZE) or, when WORLD_SIZE is unset, `python bench.py --gpus G` starts its own G rank processes (as the reference's
ParallelGroundTruthModel forks its own workers, icem/models/gt_par_model.py:26-37) and rank 0 prints the line.  The
headline of a multi-GPU line is weak scaling at the metric's population per GPU (global N = G * 4096, sharded by global
trajectory index); `also` is weak scaling at N = 65536 per GPU (the size north_star's multi-GPU target is stated on) and
`strong` is BASELINE configs[3] as written: N = 65536 GLOBAL sharded over the G GPUs (8192 per GPU at 8).  The ranks' K
candidate records per CEM iteration move through the library's own exchange (icem_exchange_*: IPC-mapped peer blocks,
peer-to-peer stores over xGMI, flags polled by the merge); where that cannot be connected, through the library's own
ncclAllGather on the launch stream (icem_allgather_elites); only if neither works, through a host-driven
torch.distributed all-gather -- every leg says which one ran (`exchange.kind`), its latency, whether the block is
fine-grained and how many waits timed out.  torch.distributed carries handles / ids once at set-up and the barriers
around the timed region, nothing inside it.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel, from HIP events recorded on
the launch stream inside the library (icem_profile_*), in a second pass over the same steps: SURVEY 8(d)'s HBM
definition (`bound`/`achieved`/`peak`/`frac`) plus the same launch priced against the f32 vector / matrix peak
(`compute`) and which of the two roofs is nearer (`binding_roof`).  `cpu_baseline` times oracle/icem_oracle.c (a
plain-C restatement, "port") on all host cores; `cpu_baselines` adds the NumPy restatement on one thread (what the
reference's main.py:23 forces) and on the host's default threads, and the C port on one thread.  `build` says which
sources the loaded library was built from.
"""
import argparse
import json
import os
import sys
import time

# the host driver only supports dmabuf IPC: without this RCCL and the exchange's hipIpc handles fail (read at HSA start-up)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # this process's own OpenMP teams (torch) sleep between uses; the CPU
                                                     # baselines run in child processes with teams of their own

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32 vector = f32 matrix peak (they share the pipe)
BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA (never the 2:1-sparsity headline)

WORKLOADS = {
    "c2": dict(N=4096, h=30, d=6, o=17, beta=0.25, iters=5, name="HalfCheetah-shaped synthetic, N=4096 h=30 d=6 o=17 beta=0.25, 5 CEM iters"),
    "c4": dict(N=65536, h=30, d=6, o=17, beta=0.25, iters=5, name="HalfCheetah-shaped synthetic, N=65536 h=30 d=6 o=17 beta=0.25, 5 CEM iters"),
    # BASELINE.json configs[2]: HumanoidStandup action shapes (d=17, bounds +-0.4, beta=2.0, 3 iterations): c3 at the real
    # env's o=378 observations (the model step is a GEMM: compute bound), c3l on a 24-dim tanh latent of
    # which obs[2] enters the cost (the HBM-side variant SURVEY 8(d) allows); not default bench lines
    # BASELINE configs[4]: learned dynamics (the declared RSSM, fused bf16-MFMA rollout), controller-driven MPC steps
    "c5": dict(N=1024, h=12, d=6, o=230, beta=0.25, iters=5, env="rssm", kind=-1,
               name="learned-dynamics (declared RSSM 200+30, GRU) N=1024 h=12 d=6 beta=0.25, 5 CEM iters, bf16 MFMA rollout"),
    "c3": dict(N=16384, h=30, d=17, o=378, beta=2.0, iters=3, env="humanoid", kind=1,
               name="HumanoidStandup-shaped synthetic at the env's real width, N=16384 h=30 d=17 o=378 beta=2.0, 3 CEM iters, dense tanh model"),
    "c3l": dict(N=16384, h=30, d=17, o=24, beta=2.0, iters=3, env="humanoid", kind=1,
                name="HumanoidStandup-shaped synthetic, N=16384 h=30 d=17 o=24 (latent) beta=2.0, 3 CEM iters, tanh model"),
    # the reference's other shipped settings (settings/{door,relocate,fpp}; beta from README.md:21-29) at the headline population:
    # cost-term envs (icem_cost_terms) on the TileHN kernel -- recorded lines under profiles/, not default bench lines
    "door": dict(N=4096, h=30, d=28, o=39, beta=2.5, iters=5, env="door", kind=1,
                 name="Door-shaped synthetic (mjenvs.py:57-78 cost terms), N=4096 h=30 d=28 o=39 beta=2.5, 5 CEM iters, tanh model"),
    "relocate": dict(N=4096, h=30, d=30, o=39, beta=3.5, iters=5, env="relocate", kind=1,
                     name="Relocate-shaped synthetic (mjenvs.py:155-174 cost terms), N=4096 h=30 d=30 o=39 beta=3.5, 5 CEM iters, tanh model"),
    "fpp": dict(N=4096, h=30, d=4, o=28, beta=3.0, iters=5, env="fpp", kind=0,
                name="FetchPickAndPlace-shaped synthetic (robotics.py:150-164 cost), N=4096 h=30 d=4 o=28 beta=3.0, 5 CEM iters, linear model"),
}


def make_env(w):
    from icem_amd import halfcheetah_env, humanoid_standup_env
    from icem_amd import envs as E
    kind = w.get("env")
    if kind == "humanoid":
        return humanoid_standup_env(w["o"])
    if kind in ("door", "relocate", "fpp"):
        return {"door": E.door_env, "relocate": E.relocate_env, "fpp": E.fetch_pick_and_place_env}[kind]()
    return halfcheetah_env(w["o"])


def make_planner(w, rank, world, seed=1234, cost_mode="sum", global_n=None, dtype="f32"):
    """One rank's planner of a run over `world` GPUs; global population = global_n (strong scaling) or w["N"] per GPU."""
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner
    env = make_env(w)
    model = DeviceSyntheticModel.make(w["o"], w["d"], kind=w.get("kind", 0))
    cfg = IcemConfig(horizon=w["h"], act_dim=w["d"], num_traj=global_n if global_n else w["N"] * world, opt_iters=w["iters"],
                     noise_beta=w["beta"], dtype=dtype, seed=seed, rank=rank, world=world, cost_mode=cost_mode)
    pl = IcemPlanner(cfg, env.action_space.low, env.action_space.high, device=f"cuda:{torch.cuda.current_device()}")
    pl.set_model(model.kind, model.A, model.B)
    pl.set_cost_spec(env.cost_spec)
    pl.reset()
    pl.obs0.copy_(torch.as_tensor(0.1 * np.random.RandomState(0).randn(w["o"]), dtype=pl.dt))
    if world > 1:
        # ICEM_BENCH_EXCHANGE = library (default: IPC peer-to-peer exchange, then the in-library RCCL all-gather, then the
        # host-driven one) | rccl (skip the first) | host (skip both)
        want = os.environ.get("ICEM_BENCH_EXCHANGE", "library")
        ok = want == "library" and pl.connect_exchange()  # the 64-byte IPC handles travel once
        if not ok and want in ("library", "rccl"):
            pl.connect_rccl()                             # the 128-byte ncclUniqueId travels once
    return pl, model, env


def run_steps(pl, n, world):
    for _ in range(n):
        pl.plan_step_resident()


def algorithmic_bytes_per_trajstep(kernel, d, h):
    """the whole loop moves 8d + 8/h bytes per traj-step in f32 (actions written once and
    read once, costs written once and read once).  A kernel is charged its own share of that."""
    return {"sample_clip": 4.0 * d, "rollout_cost": 4.0 * d + 4.0 / h, "sample_rollout": 8.0 * d + 8.0 / h}.get(kernel)


def algorithmic_flops_per_trajstep(kernel, d, h, o):
    """f32 operations the kernel's arithmetic needs per traj-step: rollout = the dense model step 2(o+d)o plus the
    cost 2d+4; sampler = per (trajectory, dim) row the folded inverse DFT 2(h^2/2 + h) plus Box-Muller / affine / clip
    ~6h, i.e. d(h + 8) per traj-step.  (The fused kernel does both.)"""
    roll = 2.0 * (o + d) * o + 2.0 * d + 4.0
    samp = d * (h + 8.0)
    return {"sample_clip": samp, "rollout_cost": roll, "sample_rollout": samp + roll}.get(kernel)


def cpu_baseline_sample(w, model, env, budget_s=4.0, samples=3):
    """(child process) oracle/icem_oracle.c (float64, OpenMP over trajectories) on whole MPC steps' worth of iterations
    of the same workload: `samples` samples of ~budget_s each, the BEST one reported (the others ride along)."""
    from oracle import c_oracle as CO
    lib = CO.load()
    cores = lib.icem_c_num_threads()
    h, d, o, K = w["h"], w["d"], w["o"], 10
    pops = []
    n = w["N"]
    for i in range(w["iters"]):
        if i:
            n = max(2 * K, int(n / 1.25))
        pops.append(n)
    A, B = CO.c64(model.A), CO.c64(model.B)
    obs0 = 0.1 * np.random.RandomState(0).randn(o)
    low, high = np.asarray(env.action_space.low, dtype=np.float64), np.asarray(env.action_space.high, dtype=np.float64)
    c = env.cost_spec
    actions = np.zeros((pops[0], h, d))
    costs = np.zeros(pops[0])
    idx = np.zeros(K, dtype=np.int32)
    ec = np.zeros(K)
    rates, reps_all, el_all = [], 0, 0.0
    for _ in range(samples):
        done, reps, t0 = 0, 0, time.perf_counter()
        while True:
            mean = np.zeros((h, d)) + (high + low) / 2
            std = np.ones((h, d)) * (high - low) / 2 * 0.5
            for it, n_it in enumerate(pops):
                lib.icem_c_iteration(n_it, h, d, o, K, w["beta"], 0.1, 1234, reps * 8 + it, 10, model.kind, A, B, obs0, low, high,
                                     c.ctrl_weight, c.lin_idx, c.lin_weight, c.flip_idx, c.flip_penalty, c.flip_thresh,
                                     mean, std, actions, costs, idx, ec)
                done += n_it * h
            reps += 1
            el = time.perf_counter() - t0
            if el > budget_s or reps >= 4096:
                break
        rates.append(done / el)
        reps_all += reps
        el_all += el
    return {"value": max(rates), "unit": "traj-steps/s", "cores": cores, "kind": "port", "samples": [round(r) for r in rates],
            "sample": f"best of {samples} samples of ~{budget_s:.0f} s ({reps_all} MPC steps of the same workload in {el_all:.1f} s, "
                      f"{sum(pops)} trajectories x h={h} each); oracle/icem_oracle.c, float64, OpenMP over trajectories, "
                      f"{cores} thread(s) bound to cores (OMP_PROC_BIND=close, OMP_PLACES=cores, active waits)"}


def usable_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def c_port_threads(w):
    """Threads the C port's parallel regions can use: one region is one population (1676..4096 rows at c2) split
    statically over the team, so beyond ~32 rows per thread the fork / join of a region costs more than its share of the
    work.  (SMT siblings do not help a float64 FMA loop: at most one thread per two logical CPUs beyond 16.)"""
    K, n = 10, w["N"]
    for i in range(1, w["iters"]):
        n = max(2 * K, int(n / 1.25))
    cpus = usable_cores()
    phys = cpus if cpus <= 16 else cpus // 2
    return max(1, min(phys, n // 32))


def cpu_baseline(workload, threads, budget_s=4.0, samples=3):
    """The C port in a CHILD process whose OpenMP team is sized and bound before libgomp starts: `threads` threads, one
    per core, spinning between the (short) parallel regions.  The parent's environment is left alone -- its own OpenMP
    teams (torch) keep sleeping between uses."""
    import subprocess
    e = dict(os.environ, HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="close", OMP_PLACES="cores",
             OMP_WAIT_POLICY="active", OMP_DYNAMIC="false")
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-c-child", workload, str(budget_s), str(samples)], env=e,
                           capture_output=True, text=True, timeout=240)
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    except Exception as ex:  # a baseline must never take the GPU numbers down with it
        return {"value": None, "unit": "traj-steps/s", "cores": threads, "kind": "port", "error": repr(ex)[:200]}


def numpy_baseline_child(workload, budget_s):
    """(child process) the NumPy restatement of the loop -- oracle/icem_oracle.py: vectorised sample -> clip -> rollout
    with the same synthetic model -> cost -> argsort[:K] -> refit, float64, np.random's legacy normals like the
    reference -- on whole MPC steps of the workload until ~budget_s; prints one JSON object."""
    from oracle import icem_oracle as O
    from icem_amd import halfcheetah_env, humanoid_standup_env
    w = WORKLOADS[workload]
    env = humanoid_standup_env(w["o"]) if w.get("env") == "humanoid" else halfcheetah_env(w["o"])
    om = O.SyntheticModel.make(w["o"], w["d"], kind=w.get("kind", 0))
    oc = O.CostSpec.humanoid_standup() if w.get("env") == "humanoid" else O.CostSpec.halfcheetah(w["o"])
    np.random.seed(0)
    orc = O.IcemOracle(O.IcemParams(horizon=w["h"], num_simulated_trajectories=w["N"], opt_iterations=w["iters"], noise_beta=w["beta"]),
                       np.asarray(env.action_space.low, dtype=np.float64), np.asarray(env.action_space.high, dtype=np.float64),
                       lambda ob, ac: O.rollout_costs(om, oc, ob, ac), lambda num: O.legacy_white_noise(num, w["d"], w["h"]))
    orc.beginning_of_rollout()
    obs = 0.1 * np.random.RandomState(0).randn(w["o"])
    pops = O.population_sizes(w["N"], 10, 1.25, w["iters"])
    reps, t0 = 0, time.perf_counter()
    while True:
        orc.get_action(obs)
        orc.trace.clear()
        reps += 1
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    print(json.dumps({"value": reps * sum(pops) * w["h"] / el, "reps": reps, "seconds": el, "traj": sum(pops)}))


def extra_cpu_baselines(workload, w, model, env):
    """the NumPy restatement with OMP_NUM_THREADS=1 (what the reference's main.py:23 forces) and with the
    host's default threads, and the C port on one thread -- each a bounded sample, in child processes so that the thread
    counts are really what the label says."""
    import subprocess
    out = []

    def child(extra_env, label, cores):
        e = dict(os.environ)
        e.update(extra_env)
        e["HIP_VISIBLE_DEVICES"] = ""
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-numpy-child", workload, "6"], env=e,
                               capture_output=True, text=True, timeout=180)
            j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            out.append({"value": j["value"], "unit": "traj-steps/s", "cores": cores, "kind": "port", "what": label,
                        "sample": f"{j['reps']} MPC step(s) ({j['traj']} trajectories x h={w['h']} each) in {j['seconds']:.1f} s; "
                                  "oracle/icem_oracle.py, NumPy float64, legacy np.random normals"})
        except Exception as ex:  # a baseline must never take the GPU numbers down with it
            out.append({"value": None, "what": label, "error": repr(ex)[:200]})
    one = {"OMP_NUM_THREADS": "1", "OPENBLAS_NUM_THREADS": "1", "MKL_NUM_THREADS": "1"}
    child(one, "NumPy restatement, 1 thread (OMP_NUM_THREADS=1 as icem/main.py:23)", 1)
    child({}, "NumPy restatement, host default threads (BLAS only; the rest of NumPy is single-threaded)", os.cpu_count())
    one_c = cpu_baseline(workload, 1, budget_s=4.0, samples=1)
    one_c["what"] = "C port (oracle/icem_oracle.c), 1 thread"
    out.append(one_c)
    return out


KERNEL_FAMILY = {"rssm_rollout": ("rssm_split_kernel", "rssm_rollout_kernel"),
                 "sample_clip": ("sample_folded_kernel", "sample_folded_merge_kernel", "noise_rows_kernel"),
                 "rollout_cost": ("rollout16_kernel", "rollout_wide_kernel", "rollout_wide_split_kernel"),
                 # one launch per iteration: the small-population kernel, or the noise-ahead launch of large populations
                 "sample_rollout": ("sample_rollout_kernel", "iter_ahead_kernel")}


def measured_traffic(kernel_class, workload):
    """HBM bytes per launch of that kernel class from the committed rocprofv3 PMC passes over this same command
    (profiles/rNN_<workload>_hbm_traffic.json, written by tools/summarize_profile.py: separate --pmc FETCH_SIZE /
    WRITE_SIZE runs, reads with the gfx950 x2 correction), and the average duration the kernel trace of that
    command shows for it.  None if no such profile is in the tree."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload}_hbm_traffic.json")))
    if not files:
        return None, None, None
    tab = json.load(open(files[-1]))
    tot, n, dur = 0.0, 0, 0.0
    for name in KERNEL_FAMILY.get(kernel_class, ()):
        if name in tab:
            tot += (tab[name]["read_bytes_per_launch"] + tab[name]["write_bytes_per_launch"]) * tab[name]["launches"]
            dur += (tab[name].get("trace_avg_duration_ns") or 0.0) * tab[name]["launches"]
            n += tab[name]["launches"]
    return (tot / n, os.path.relpath(files[-1], ROOT), dur / n * 1e-3 or None) if n else (None, None, None)


def event_bracket_us(pl):
    """The share of a ``profile_read`` span that is the event pair and the launch's dispatch, not the kernel: an event pair
    around a one-wave kernel that spins for 15 us of the wall clock, minus the time that kernel reports it ran
    (``icem_profile_overhead``; medians of 200, on the launch stream).  Subtracted from every span so that the per-kernel
    times are the ones the launches contribute to the step they were taken from (a time that sums to more than ms_per_step
    is not evidence of anything).  The in-kernel time leaves out the wave's launch and retirement (part of a kernel's
    duration in a rocprofv3 trace), so the corrected times sit below the trace's by that much; the committed trace's own
    figure rides along as ``rocprof_trace_avg_us``."""
    out = {}
    for spin in (15.0, 40.0):   # two durations: the bracket must not depend on the kernel's length
        pair, kern = pl.profile_overhead(200, spin)
        out[f"spin_{int(spin)}us"] = {"pair_us": round(pair, 3), "kernel_us": round(kern, 3), "bracket_us": round(pair - kern, 3)}
    out["subtracted_us"] = round(max(0.0, min(v["bracket_us"] for v in out.values())), 3)
    return out


def without_bracket(prof, bracket):
    """{kernel: (ms, launches, units)} with the event bracket taken out of every span (the bracket is the SMALLEST one the
    calibration kernel showed, so a span cannot be shorter; a class that came out non-positive is dropped by roofline_of)."""
    sub = bracket["subtracted_us"] * 1e-3
    return {k: (max(ms - sub * n, 0.0), n, u) for k, (ms, n, u) in prof.items()}


def fit_to_step(prof, profiled_steps, ms_per_step):
    """Price every kernel on its SHARE OF THE TIMED STEP: the bracket-corrected event times are scaled UP so that they sum
    to ms_per_step exactly -- launch gaps are charged to the kernels pro rata (conservative: a roofline fraction computed
    from these can only be lower than the kernel alone would show).  Times that sum to more than 1.05 x the step are left
    as they are and flagged (`fits` false): the line then carries `roofline.valid` = false instead of fractions inflated
    by a scale-down.  Returns the profile and the record of what was done."""
    per_step_ms = sum(ms for ms, _, _ in prof.values()) / max(1, profiled_steps)
    ratio = per_step_ms / ms_per_step if ms_per_step > 0 else float("inf")
    fit = {"sum_of_kernel_times_ms_per_step": per_step_ms, "ms_per_step": ms_per_step, "ratio": ratio, "fits": ratio <= 1.05,
           "priced_on": "share of the timed step (event time x ms_per_step / sum of event times)"}
    if ratio > 1.05:
        # event times that exceed the step they were taken from are not evidence of anything: they are NOT scaled down (that
        # would inflate every fraction computed from them) -- the caller prints the line with `roofline.valid` false
        print(f"bench.py: per-kernel event times sum to {ratio:.2f} x the timed step: roofline fractions marked invalid", file=sys.stderr)
        fit["priced_on"] = "raw event times (they exceed the timed step: fractions invalid)"
        return dict(prof), fit
    return {k: (ms / ratio, n, u) for k, (ms, n, u) in prof.items()}, fit


def attainable(w, units, launches, tile_arith, kernel):
    """The floor of ONE launch of `kernel` on this chip at its own operation count: max(HBM floor at the algorithmic bytes,
    ALU floor).  ALU floor = the model step's multiply-adds on the pipe they run on -- the exact-f32 pipe (vector and
    v_mfma_f32_16x16x4_f32 share it: 157.3 TFLOP/s), or, in fp16 planes, THREE fp16 products per multiply-add at the dense
    fp16 peak -- plus the sampler's and the cost's f32 vector operations at the f32 peak (they do not overlap the matrix
    pipe's issue slot on gfx950: one VALU/MFMA issue per cycle and SIMD)."""
    d, h, o = w["d"], w["h"], w["o"]
    per = units / launches
    model = 2.0 * (o + d) * o
    rest = algorithmic_flops_per_trajstep(kernel, d, h, o) - (model if kernel != "sample_clip" else 0.0)
    hbm_us = per * algorithmic_bytes_per_trajstep(kernel, d, h) / (HBM_PEAK_GBS * 1e9) * 1e6
    if kernel == "sample_clip":
        alu_us = per * rest / (F32_PEAK_TFLOPS * 1e12) * 1e6
    elif tile_arith:
        alu_us = per * (3.0 * model / (BF16_PEAK_TFLOPS * 1e12) + rest / (F32_PEAK_TFLOPS * 1e12)) * 1e6
    else:
        alu_us = per * (model + rest) / (F32_PEAK_TFLOPS * 1e12) * 1e6
    return {"hbm_floor_us": hbm_us, "alu_floor_us": alu_us, "attainable_us": max(hbm_us, alu_us),
            "model_step_arithmetic": "two fp16 planes, 3 products per multiply-add on v_mfma_f32_16x16x32_f16" if tile_arith
                                     else "exact f32 (v_mfma_f32_16x16x4_f32 / v_fmac_f32: one shared pipe)"}


def roofline_of(prof, w, workload=None, fit=None, tile_arith=0):
    prof = {k: v for k, v in prof.items() if v[0] > 0}
    if not prof:
        return None
    dom = max(prof, key=lambda k: prof[k][0])
    ms, launches, units = prof[dom]
    bpu = algorithmic_bytes_per_trajstep(dom, w["d"], w["h"])
    if bpu is None or ms <= 0:
        return None
    valid = bool(fit["fits"]) if fit else True
    achieved = units * bpu / (ms * 1e-3) / 1e9
    fpu = algorithmic_flops_per_trajstep(dom, w["d"], w["h"], w["o"])
    tflops = units * fpu / (ms * 1e-3) / 1e12
    compute = {"achieved": tflops, "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / F32_PEAK_TFLOPS,
               "flops_per_traj_step": fpu, "flops_per_launch": units * fpu / launches,
               "note": "algorithmic f32 operations over the EXACT-f32 peak (f32 vector and f32 MFMA share one pipe on gfx950: 155 TF "
                       "measured); in fp16 planes the model step's share runs on the 16-bit matrix cores -- see attainable"}
    traffic, src, trace_us = measured_traffic(dom, workload) if workload else (None, None, None)
    # avg_launch_us: HIP events around every launch on the launch stream, the calibrated event bracket taken out
    # (event_bracket_us), then priced as the launch's share of the timed step (fit_to_step: the times sum to ms_per_step);
    # rocprof_trace_avg_us: the committed kernel trace's figure for the same kernels, for comparison
    if w["o"] > 32 and dom == "rollout_cost":
        # wide observations: the rollout is a GEMM three orders of magnitude above the ridge ("declare that
        # stage compute-bound").  It runs on the fp16 matrix cores with every f32 operand x 2^k split in two fp16 planes:
        # THREE fp16 products per algorithmic multiply-add (k_rollout_wide_split.hip).  `frac` is the ALGORITHMIC rate
        # (one multiply-add = 2 flop, whatever executes it) over the peak of the pipe used; the executed products'
        # utilisation of that pipe and the algorithmic rate over the exact-f32 matrix peak ride beside it.
        executed = 3.0 * tflops
        return {"bound": "mfma", "achieved": tflops, "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / BF16_PEAK_TFLOPS,
                "valid": valid, "what": "algorithmic flops per launch / launch time / dense fp16 peak (the pipe the kernel runs on)",
                "executed_TFLOPs": executed, "executed_frac": executed / BF16_PEAK_TFLOPS,
                "vs_exact_f32_matrix_peak": tflops / F32_PEAK_TFLOPS,
                "traffic": traffic, "traffic_source": src, "kernel": dom, "avg_launch_us": 1e3 * ms / launches,
                "rocprof_trace_avg_us": trace_us, "launches": launches, "flops_per_traj_step": fpu,
                "algorithmic_flops_per_launch": units * fpu / launches,
                "products_per_multiply_add": 3,
                "dtype": "2-way fp16 split of f32 operands x 2^k, three products per multiply-add, f32 accumulation (v_mfma_f32_16x16x32_f16)",
                "hbm": {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "bytes_per_traj_step": bpu}, "binding_roof": "fp16-mfma"}
    att = attainable(w, units, launches, tile_arith, dom)
    att["frac_of_attainable"] = att["attainable_us"] / (1e3 * ms / launches)
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "valid": valid, "traffic": traffic, "traffic_source": src, "kernel": dom, "avg_launch_us": 1e3 * ms / launches,
            "rocprof_trace_avg_us": trace_us, "launches": launches,
            "algorithmic_bytes_per_launch": units * bpu / launches, "bytes_per_traj_step": bpu,
            "compute": compute, "attainable": att,
            "binding_roof": "hbm" if att["hbm_floor_us"] >= att["alu_floor_us"] else ("fp16-mfma + f32-alu" if tile_arith else "f32-alu")}


def distribution_checksum(pl):
    """A float64-exact checksum of the bits of a planner's mean | std (low 24 bits of every element's image, summed)."""
    mean, std = getattr(pl, "mean", None), getattr(pl, "std", None)
    if mean is None or std is None:
        return 0.0
    x = torch.cat([torch.as_tensor(mean).flatten(), torch.as_tensor(std).flatten()]).contiguous()
    bits = x.view(torch.int32 if x.element_size() == 4 else torch.int64).to(torch.int64)
    return float((bits & 0xFFFFFF).sum().item())


class RecordsPathFailed(RuntimeError):
    """Some rank's launches reported a failure of the path the ranks' elite records travel on (agreed on by all ranks)."""


def timed_steps(pl, steps, warmup, world, spread=None):
    """W untimed steps, then K timed ones between barrier + synchronize pairs; the MAX over ranks.  That first block of
    exactly K steps is what `value` is computed from.  A block shorter than 10 ms says little about a box (clock ramps, a
    neighbour's interrupt): when `spread` (a dict) is given, further blocks of K steps are timed the same way until 10 ms
    have been covered, and their min / median / max ms per step are recorded in it.
    world > 1: a rank whose step raises (a bounded wait for a peer's records ran out) does NOT leave the collective
    pattern -- every block ends in the same all-reduce, which carries the ranks' failure flags; if any is set, every rank
    raises RecordsPathFailed and the caller degrades the path on all of them (timed_with_fallback)."""
    from icem_amd import _lib as L
    on_gpu = torch.cuda.is_available()   # (the CPU test of the ranks' agreement drives this loop with a stand-in planner)

    def sync():
        if on_gpu:
            torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            if on_gpu:
                torch.cuda.synchronize()

    def block(n):
        sync()
        failed = 0.0
        t0 = time.perf_counter()
        try:
            run_steps(pl, n, world)
        except L.IcemError as ex:
            # Only the in-library exchange has no collective INSIDE a step: a rank may leave the block early and meet its peers at
            # the all-reduce below.  On the RCCL / host-driven paths the peers are inside a step's all-gather: leaving it would
            # deadlock them -- there the error ends this rank (and with it the run: the launcher takes the others down).
            safe = getattr(pl, "leaves_block_safely", None)   # (the CPU test's stand-in planner says so itself)
            if world > 1 and not (bool(getattr(pl, "_exchange", False)) if safe is None else safe):
                raise
            failed = 1.0
            print(f"bench.py: rank {pl.cfg.rank}: {ex}", file=sys.stderr)
        sync()
        el = time.perf_counter() - t0
        if world > 1 and getattr(pl, "_exchange", False):
            # the steps are enqueued long before the device runs them: a wait that ran out in THIS block has raised nothing
            # yet -- read (and clear) the status word the kernels report into
            status = pl.exchange_status()[0]
            if status & 1:
                pl._xchg_status_seen = status
                failed = 1.0
        if world > 1:
            import torch.distributed as dist
            # every rank refits the same distribution from the same records: a checksum of its bits rides in the same
            # all-reduce (max of h and of -h), so records that arrive stale or torn on some rank -- nothing raises then --
            # end the block like a failure of the path
            h = distribution_checksum(pl)
            t = torch.tensor([el, failed, h, -h], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el, failed = float(t[0].item()), float(t[1].item())
            if not failed and float(t[2].item()) != -float(t[3].item()):
                failed = 1.0
                pl._degrade_reason = "the ranks' distributions differed after a timed block (records stale or torn on some rank)"
                if spread is not None:
                    spread["_ranks_disagreed"] = spread.get("_ranks_disagreed", 0) + 1
                if pl.cfg.rank == 0:
                    print("bench.py: the ranks' distributions differ after a block (records stale or torn on some rank)", file=sys.stderr)
        if failed:
            raise RecordsPathFailed()
        return el
    block(warmup)
    el = block(steps)
    if spread is not None:
        blocks = [el]
        while sum(blocks) < 10e-3 and len(blocks) < 64:   # (el is the max over ranks: every rank takes the same number)
            blocks.append(block(steps))
        per = sorted(1e3 * b / steps for b in blocks)
        spread.update({"blocks_of_k_steps": len(blocks), "covered_ms": 1e3 * sum(blocks), "ms_per_step_first_block": 1e3 * el / steps,
                       "ms_per_step_min": per[0], "ms_per_step_median": per[len(per) // 2], "ms_per_step_max": per[-1]})
    return el


def timed_with_fallback(pl, steps, warmup, world, spread=None):
    """timed_steps; where the records' path fails at run time, all ranks step down together (IcemPlanner.degrade_exchange:
    in-library exchange -> in-library RCCL all-gather -> host-driven all-gather) and the measurement starts over."""
    degraded, disagreed = [], 0
    for _ in range(3):
        try:
            if spread is not None:
                spread.clear()
            el = timed_steps(pl, steps, warmup, world, spread)
            if spread is not None:
                if degraded:
                    spread["records_path_degraded_to"] = degraded
                if world > 1:
                    spread["ranks_agree_after_every_block"] = True   # (a block that ended otherwise raised)
                    if disagreed:
                        spread["blocks_the_ranks_disagreed_after"] = disagreed
            return el
        except RecordsPathFailed:
            disagreed += (spread or {}).get("_ranks_disagreed", 0)
            degraded.append(pl.degrade_exchange())
            if pl.cfg.rank == 0:
                print(f"bench.py: the records' path failed at run time; all ranks now on: {degraded[-1]}", file=sys.stderr)
    raise SystemExit("bench.py: the ranks' elite records do not get through on any path")


def loop_floor_ms(w, pops_total, tile_arith):
    """Whole MPC step: max(HBM floor at 8d + 8/h bytes per traj-step, ALU floor at the loop's own operation count) --
    see attainable()."""
    units = pops_total * w["h"]
    a = attainable(w, units, 1, tile_arith, "sample_rollout")
    return {"hbm_floor_ms": a["hbm_floor_us"] * 1e-3, "alu_floor_ms": a["alu_floor_us"] * 1e-3, "attainable_ms": a["attainable_us"] * 1e-3,
            "model_step_arithmetic": a["model_step_arithmetic"]}


def measure_also(name, rank=0, world=1, steps=200, warmup=20, global_n=None):
    """The large-population configuration the north-star targets are stated on (N=65536 per GPU), measured in the same
    run (every rank takes part when world > 1: weak scaling at 65536 rows per GPU; global_n: strong scaling, that many
    rows over all GPUs): whole-loop traj-steps/s, ms per MPC step, dominant-kernel roofline, and the whole loop's
    algorithmic bytes (8d+8/h per traj-step) over the step time."""
    w = WORKLOADS[name]
    pl, _, _ = make_planner(w, rank, world, global_n=global_n)
    spread = {}
    el = timed_with_fallback(pl, steps, warmup, world, spread)
    pl.profile_enable(True)
    run_steps(pl, 10, world)
    torch.cuda.synchronize()
    raw = pl.profile_read()
    pl.profile_enable(False)
    bracket = event_bracket_us(pl)
    prof, fit = fit_to_step(without_bracket(raw, bracket), 10, 1e3 * el / steps)
    ts = sum(pl.population_sizes) * w["h"]  # global
    loop_bytes = ts * (8.0 * w["d"] + 8.0 / w["h"])
    n_glob = global_n if global_n else w["N"] * world
    out = {"workload": w["name"] if not global_n else w["name"].replace(f"N={w['N']}", f"N={n_glob} global"),
           "scaling": "strong" if global_n else "weak", "per_gpu_population": -(-n_glob // world), "global_population": n_glob,
           "n_gpus": world, "value": ts * steps / el, "unit": "traj-steps/s", "ms_per_mpc_step": 1e3 * el / steps,
           "timed_region": spread,
           "roofline": roofline_of(prof, w, name if world == 1 else None, fit, pl.tile_arith if w["o"] <= 32 else 0),
           "kernels_us": {k: round(1e3 * v[0] / v[1], 2) for k, v in prof.items()},
           "kernels_us_events": {k: round(1e3 * v[0] / v[1], 2) for k, v in without_bracket(raw, bracket).items()},
           "kernels_us_with_event_bracket": {k: round(1e3 * v[0] / v[1], 2) for k, v in raw.items()},
           "event_bracket": bracket, "kernel_times_vs_step": fit,
           "whole_loop_algorithmic_GBps": loop_bytes * steps / el / 1e9,
           "whole_loop_frac_of_hbm_peak": loop_bytes * steps / el / 1e9 / (HBM_PEAK_GBS * world)}
    if w["o"] <= 32:
        out["attainable"] = loop_floor_ms(w, sum(pl.population_sizes) / world, pl.tile_arith)
        out["attainable"]["frac_of_attainable"] = out["attainable"]["attainable_ms"] / out["ms_per_mpc_step"]
    if world > 1:
        out["exchange"] = exchange_report(pl)
    return out


def measure_batched(name="c2", batches=(2, 4, 8, 16), steps=100, warmup=12, solo_ms=None):
    """B independent planners of the headline configuration (different models' seeds, costs and observations; the reference's
    parallel episodes, icem/misc/rollout_utils.py:46-58, 129-152) advanced by icem_plan_step_batch: every stage one launch for
    all of them.  Per B: ms per (batched) MPC step, aggregate traj-steps/s over all problems, the whole loop's algorithmic
    bytes over the step time as a fraction of the HBM line, and the aggregate relative to B solo steps of this run."""
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner
    w = WORKLOADS[name]
    env = make_env(w)
    out = {"workload": w["name"], "what": "B independent planners, one launch per stage (icem_plan_step_batch)", "by_B": {}}
    for B in batches:
        pls = []
        for i in range(B):
            model = DeviceSyntheticModel.make(w["o"], w["d"], kind=w.get("kind", 0), seed_a=2 * i, seed_b=2 * i + 1)
            cfg = IcemConfig(horizon=w["h"], act_dim=w["d"], num_traj=w["N"], opt_iters=w["iters"], noise_beta=w["beta"],
                             dtype="f32", seed=1234 + i)
            pl = IcemPlanner(cfg, env.action_space.low, env.action_space.high, device=f"cuda:{torch.cuda.current_device()}")
            pl.set_model(model.kind, model.A, model.B)
            pl.set_cost_spec(env.cost_spec)
            pl.reset()
            pl.obs0.copy_(torch.as_tensor(0.1 * np.random.RandomState(i).randn(w["o"]), dtype=pl.dt))
            pls.append(pl)
        for _ in range(warmup):
            IcemPlanner.plan_step_batch(pls)
        torch.cuda.synchronize()
        up0 = pls[0].batch_uploads
        t0 = time.perf_counter()
        for _ in range(steps):
            IcemPlanner.plan_step_batch(pls)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ts = B * sum(pls[0].population_sizes) * w["h"]
        loop_bytes = ts * (8.0 * w["d"] + 8.0 / w["h"])
        row = {"ms_per_batched_mpc_step": 1e3 * el / steps, "ms_per_problem_step": 1e3 * el / steps / B,
               "value": ts * steps / el, "unit": "traj-steps/s (all problems)",
               "whole_loop_algorithmic_GBps": loop_bytes * steps / el / 1e9,
               "whole_loop_frac_of_hbm_peak": loop_bytes * steps / el / 1e9 / HBM_PEAK_GBS,
               "argument_uploads_in_timed_steps": pls[0].batch_uploads - up0}
        if solo_ms:
            row["aggregate_vs_solo_steps"] = B * solo_ms / (1e3 * el / steps)
        out["by_B"][str(B)] = row
        del pls
    return out


def measure_f64(name="c2", steps=40, warmup=4):
    """The reference's own arithmetic (float64 throughout, icem.py) on the device: the strict-parity mode of the same
    workload (generic kernels, one per stage: sample_clip / rollout_cost / top-K partial + final / gather + refit) -- the
    mode the golden fixtures of tests/golden are replayed in at 1e-10.  One timed number per round so that it is seen."""
    w = WORKLOADS[name]
    pl, _, _ = make_planner(w, 0, 1, dtype="f64")
    el = timed_steps(pl, steps, warmup, 1)
    ts = sum(pl.population_sizes) * w["h"]
    return {"workload": w["name"], "dtype": "f64", "kernels": "generic_kernels.hip (strict-parity mode)", "steps": steps,
            "value": ts * steps / el, "unit": "traj-steps/s", "ms_per_mpc_step": 1e3 * el / steps,
            "algorithmic_GBps": ts * 2 * (8.0 * w["d"] + 8.0 / w["h"]) * steps / el / 1e9}


def exchange_report(pl):
    """How the ranks' elite records travelled in this run, and what one exchange costs (probe: 200 back-to-back
    exchanges inside one launch per rank, no launches around them)."""
    if not getattr(pl, "_exchange", False):
        import torch.distributed as dist
        if getattr(pl, "_rccl", False):
            # in-library fallback: ncclAllGather on the launch stream; latency of one such call (200 back to back)
            import ctypes as C
            from icem_amd import _lib as L
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(200):
                L.check(pl.lib.icem_allgather_elites(pl._h, C.c_void_p(pl.records.data_ptr()), pl._stream()))
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / 200 * 1e6
            return {"kind": "in-library: ncclAllGather of the K records on the launch stream (csrc/collective.hip, icem_allgather_elites)",
                    "ran": "rccl", "latency_us": us, "records_bytes_per_rank": pl.K * (pl.h * pl.d + 2) * 4, "finegrained_block": None,
                    "timeouts": 0, "host_collectives_in_timed_loop": 0, "rccl_library": pl.lib.icem_rccl_library().decode(),
                    "in_library_exchange_error": getattr(pl, "exchange_error", None)}
        return {"kind": "torch.distributed all_gather_into_tensor per CEM iteration (host-driven)", "ran": "host", "latency_us": None,
                "finegrained_block": None, "timeouts": None, "host_collectives_in_timed_loop": pl.cfg.opt_iters,
                "in_library_exchange_error": getattr(pl, "exchange_error", None), "rccl_error": getattr(pl, "rccl_error", None)}
    us = pl.exchange_probe(200)
    status, fine = pl.exchange_status()
    return {"kind": "in-library: P2P stores into IPC-mapped peer blocks + flags (csrc/exchange.hip)", "ran": "ipc", "latency_us": us,
            "records_bytes_per_rank": pl.K * (pl.h * pl.d + 2) * 4, "finegrained_block": fine, "timeouts": status,
            "host_collectives_in_timed_loop": 0}


def measure_c5(w, steps, warmup, cpu=True):
    """--workload c5: one step = MpcICemHip.get_action with DeviceRSSMModel (sampling / top-K / refit kernels + one fused
    RSSM rollout launch per CEM iteration); roofline bound = the bf16 matrix cores for the rollout kernel; CPU baseline =
    oracle/rssm_oracle.py (NumPy, float64) on the same populations."""
    from icem_amd import DeviceRSSMModel, MpcICemHip, halfcheetah_env
    env = halfcheetah_env(17)   # action space only (d=6, +-1)
    model = DeviceRSSMModel(seed=3)
    ctrl = MpcICemHip(env=env, forward_model=model, horizon=w["h"], num_simulated_trajectories=w["N"], factor_decrease_num=1.25,
                      cost_along_trajectory="sum", dtype="f32", seed=1234,
                      action_sampler_params=dict(alpha=0.1, elites_size=10, opt_iterations=w["iters"], init_std=0.5,
                                                 use_mean_actions=True, keep_previous_elites=True, shift_elites_over_time=True,
                                                 fraction_elites_reused=0.3, noise_beta=w["beta"]))
    obs = 0.3 * np.random.RandomState(0).randn(w["o"])
    ctrl.beginning_of_rollout(observation=obs, state=None, mode="train")
    for _ in range(warmup):
        ctrl.get_action(obs, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctrl.get_action(obs, None)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    pops = ctrl.planner.population_sizes
    ts = sum(pops) * w["h"]
    # second pass: the rollout kernel alone, HIP events on the stream it is launched on (torch's current stream)
    orig, spans = model.rollout_cost, []

    def timed(o, a, mode=0):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(o, a, mode)
        e1.record()
        spans.append((e0, e1, a.shape[0]))
        return out
    model.rollout_cost = timed
    for _ in range(min(steps, 50)):
        ctrl.get_action(obs, None)
    torch.cuda.synchronize()
    model.rollout_cost = orig
    ms = sum(e0.elapsed_time(e1) for e0, e1, _ in spans)
    # algorithmic work of a trajectory: the reward head on the h states its steps start from, the transition h - 1 times
    # (the state behind the last step is never scored; the split launch does not compute it)
    mac_rew = sum(p.numel() for k, p in model.reference.named_parameters() if p.ndim == 2 and k.startswith("rew"))
    mac_tr = sum(p.numel() for k, p in model.reference.named_parameters() if p.ndim == 2 and not k.startswith("rew"))
    macs = (w["h"] * mac_rew + (w["h"] - 1) * mac_tr) / w["h"]
    flops = 2.0 * macs * w["h"] * sum(n for _, _, n in spans)
    achieved = flops / (ms * 1e-3) / 1e12
    traffic, src, trace_us = measured_traffic("rssm_rollout", "c5")
    roofline = {"bound": "mfma", "achieved": achieved, "peak": 2500.0, "unit": "TFLOP/s", "frac": achieved / 2500.0, "traffic": traffic,
                "traffic_source": src, "rocprof_trace_avg_us": trace_us,
                "kernel": "rssm_split_kernel" if max(n for _, _, n in spans) <= 65536 else "rssm_rollout_kernel", "avg_launch_us": 1e3 * ms / len(spans), "launches": len(spans),
                "algorithmic_flops_per_traj_step": 2.0 * macs, "dtype": "bf16 operands, f32 accumulation"}
    out = {"metric": "traj-steps/sec (N x h), iCEM inner planning loop", "value": ts * steps / elapsed, "unit": "traj-steps/s",
           "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": w["name"], "per_gpu_population": w["N"], "traj_per_mpc_step": sum(pops),
                      "model": "declared RSSM, random weights (icem_amd.models.declared_rssm)", "cost": "-reward head",
                      "rng": "Philox4x32-10-seeded xoshiro128++ + Box-Muller", "parallelism": "n-shard x1"},
           "roofline": roofline, "cpu_baseline": None}
    if cpu:
        from oracle import rssm_oracle as RO
        P = RO.params_from_state_dict(model.reference.state_dict())
        rs = np.random.RandomState(1)
        done, reps, t0 = 0, 0, time.perf_counter()
        while time.perf_counter() - t0 < 12.0:
            for n_it in pops:
                RO.rollout_costs(P, obs, rs.uniform(-1, 1, (n_it, w["h"], w["d"])))
                done += n_it * w["h"]
            reps += 1
        el = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": done / el, "unit": "traj-steps/s", "cores": os.cpu_count(), "kind": "port",
                               "sample": f"{reps} MPC step(s) worth of rollouts ({sum(pops)} trajectories x h={w['h']} each) in {el:.1f} s; "
                                         "oracle/rssm_oracle.py, NumPy float64 (BLAS threads as configured on the host)"}
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes of this command line (the
    environment torch.distributed.run would give them: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), pass rank 0's output
    through, fail if any rank fails.  Reference analogue: ParallelGroundTruthModel forks its own workers
    (icem/models/gt_par_model.py:26-37)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    n_dev = max(1, torch.cuda.device_count())
    per_dev = -(-n // n_dev)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count if (per_dev > 1 and torch.cuda.is_available()) else 0
    for r in range(n):
        e = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ICEM_BENCH_LAUNCHER="bench.py (self-launched ranks)")
        if n_cu and "HSA_CU_MASK" not in e:
            # more ranks than GPUs (a debugging run): the ranks of a GPU each get their own slice of its CUs -- a rank's
            # launch spins on its peers' records, and a peer that cannot get a CU meanwhile is a wait for the poll budget
            # (on a node every rank has a GPU of its own and nothing is masked)
            cus = n_cu // per_dev
            e["HSA_CU_MASK"] = f"{r % n_dev}:{(r // n_dev) * cus}-{(r // n_dev + 1) * cus - 1}"
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    if rc:
        sys.exit(rc)


def dtype_name(tile_arith):
    """The arithmetic the path computes in, by name: storage is f32 everywhere; the model step of the
    tile kernels runs either as an exact f32 fmaf chain or on two fp16 planes per operand with f32 accumulation."""
    return "f32 storage; model step 2xfp16 planes (3 products), f32 accumulate" if tile_arith else "f32"


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--cpu-numpy-child":   # children of extra_cpu_baselines
        return numpy_baseline_child(sys.argv[2], float(sys.argv[3]))
    if len(sys.argv) >= 4 and sys.argv[1] == "--cpu-c-child":
        from icem_amd import DeviceSyntheticModel, halfcheetah_env, humanoid_standup_env
        w = WORKLOADS[sys.argv[2]]
        env = humanoid_standup_env(w["o"]) if w.get("env") == "humanoid" else halfcheetah_env(w["o"])
        model = DeviceSyntheticModel.make(w["o"], w["d"], kind=w.get("kind", 0))
        print(json.dumps(cpu_baseline_sample(w, model, env, budget_s=float(sys.argv[3]),
                                             samples=int(sys.argv[4]) if len(sys.argv) > 4 else 1)))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS), help="c2 = the metric's configuration (default)")
    ap.add_argument("--cost-mode", default="sum", choices=["sum", "best", "final"],
                    help="cost_along_trajectory (abstract_controller.py:82-87); the metric is quoted on 'sum'")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the extra measurements (c4 per GPU; c5 on one GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # nobody launched the ranks for us: start them ourselves (one process per GPU, this command line in each)
        return self_launch(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: running with the launcher's {world} rank(s)", file=sys.stderr)
    if os.environ.get("ICEM_BENCH_DRYRUN"):   # CPU-side test of the launch plumbing (tests/test_distributed_gloo.py): no GPU work
        import torch.distributed as dist
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            t = torch.tensor([float(rank + 1)], dtype=torch.float64)
            dist.all_reduce(t)
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dryrun": True, "n_gpus": world, "rank_sum": float(t.item()) if world > 1 else 1.0,
                              "launched_by": os.environ.get("ICEM_BENCH_LAUNCHER", "direct")}), flush=True)
        return
    n_dev = max(1, torch.cuda.device_count())
    shared_gpu = world > n_dev                        # more ranks than GPUs: a debugging run, ranks share devices
    local_rank %= n_dev
    torch.cuda.set_device(local_rank)
    backend = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL wants one GPU per rank; ranks that share a GPU rendezvous over gloo (set-up traffic and barriers only)
        backend = os.environ.get("ICEM_BENCH_BACKEND", "gloo" if shared_gpu else "nccl")
        dist.init_process_group(backend, rank=rank, world_size=world,
                                **({"device_id": torch.device(f"cuda:{local_rank}")} if backend == "nccl" else {}))

    import __graft_entry__ as ge
    from icem_amd import build as B
    stale_before = B.build_info()["stale"]
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    build = dict(B.build_info(), rebuilt_in_this_run=bool(stale_before))
    # development options: the library reads no environment variable; this tool maps ICEM_<NAME> onto icem_set_option
    from icem_amd import _lib as _L
    build["options_from_environment"] = _L.apply_env_options()

    w = WORKLOADS[args.workload]
    if args.workload == "c5":
        if world != 1:
            sys.exit("--workload c5 is a single-GPU configuration")
        out = measure_c5(w, args.steps, args.warmup, cpu=not args.no_cpu_baseline)
        out["build"] = build
        print(json.dumps(out), flush=True)
        return
    pl, model, env = make_planner(w, rank, world, cost_mode=args.cost_mode)
    per_step_trajsteps = sum(pl.population_sizes) * w["h"]  # global (all ranks) traj-steps per MPC step
    spread = {}
    elapsed = timed_with_fallback(pl, args.steps, args.warmup, world, spread)
    tile_arith = pl.tile_arith if w["o"] <= 32 else 0

    # second pass: per-kernel durations from HIP events on the launch stream
    pl.profile_enable(True)
    n_prof = min(args.steps, 50)
    run_steps(pl, n_prof, world)
    torch.cuda.synchronize()
    raw_prof = pl.profile_read()
    pl.profile_enable(False)
    bracket = event_bracket_us(pl)
    prof, fit = fit_to_step(without_bracket(raw_prof, bracket), n_prof, 1e3 * elapsed / args.steps)
    exchange = exchange_report(pl) if world > 1 else None   # collective: every rank
    also = strong = None
    if args.workload == "c2" and not args.no_also:
        del pl
        # (ranks that share ONE GPU cannot run 65536 rows each: a rank's launch fills the chip and spins on a peer that
        #  cannot get a CU until the bounded waits give up -- DESIGN section 6; the debugging run skips that leg)
        if not shared_gpu:
            also = measure_also("c4", rank, world)          # collective when world > 1
        if world > 1:   # BASELINE configs[3] as written: N = 65536 global, sharded (8192 per GPU at 8)
            # (ranks sharing one GPU: a population both ranks' launches fit the chip with, for the code path only)
            strong = measure_also("c4", rank, world, global_n=WORKLOADS["c4"]["N"] if not shared_gpu else 2048 * world)
            strong["debug_population"] = shared_gpu

    if rank == 0:
        roofline = roofline_of(prof, w, args.workload if world == 1 else None, fit, tile_arith)
        out = {
            "metric": "traj-steps/sec (N x h), iCEM inner planning loop", "value": per_step_trajsteps * args.steps / elapsed,
            "unit": "traj-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "launched_by": os.environ.get("ICEM_BENCH_LAUNCHER", "torch.distributed.run" if world > 1 else "direct"),
            "rendezvous_backend": backend, "ranks_share_a_gpu": shared_gpu,
            "vs_baseline": None, "dtype": dtype_name(tile_arith), "data": "synthetic",
            "config": {"workload": w["name"], "per_gpu_population": w["N"], "global_population": w["N"] * world,
                       "traj_per_mpc_step": per_step_trajsteps // w["h"],
                       "model": "o' = tanh(o.A + a.B) dense (synthetic)" if model.kind == 1 else "o' = o.A + a.B dense linear (synthetic)",
                       "cost": {"humanoid": "HumanoidStandup cost_fn", "door": "Door cost terms", "relocate": "Relocate cost terms",
                     "fpp": "FetchPickAndPlace cost terms"}.get(w.get("env"), "HalfCheetah cost_fn"), "cost_along_trajectory": args.cost_mode, "rng": "Philox4x32-10-seeded xoshiro128++ + Box-Muller", "parallelism": f"n-shard x{world}"},
            "ms_per_mpc_step": 1e3 * elapsed / args.steps, "timed_region": spread,
            "roofline": roofline,
            "kernels_us": {k: round(1e3 * v[0] / v[1], 2) for k, v in prof.items()},
            "kernels_us_events": {k: round(1e3 * v[0] / v[1], 2) for k, v in without_bracket(raw_prof, bracket).items()},
            "kernels_us_with_event_bracket": {k: round(1e3 * v[0] / v[1], 2) for k, v in raw_prof.items()},
            "event_bracket": bracket, "kernel_times_vs_step": fit,
            "build": build,
        }
        if w["o"] <= 32:
            out["attainable"] = loop_floor_ms(w, per_step_trajsteps / w["h"] / world, tile_arith)
            out["attainable"]["frac_of_attainable"] = out["attainable"]["attainable_ms"] / out["ms_per_step"]
        if exchange is not None:
            out["exchange"] = exchange
        # GPU measurements first: the OpenMP team of the CPU baseline keeps spinning on the host cores for a
        # while after its last parallel region and would slow down kernel launches
        if also is not None:
            out["also"] = also
        if strong is not None:
            out["strong"] = strong
        if world == 1 and args.workload == "c2" and not args.no_also:
            try:   # BASELINE configs[2] at the env's real width o = 378 (the f32 GEMM rollout): 30 MPC steps
                out["also_c3"] = measure_also("c3", 0, 1, steps=30, warmup=3)
            except Exception as ex:
                out["also_c3"] = {"error": repr(ex)[:300]}
            try:   # B independent planners of the headline configuration in one launch per stage
                out["also_batched"] = measure_batched("c2", solo_ms=out["ms_per_step"])
            except Exception as ex:
                out["also_batched"] = {"error": repr(ex)[:300]}
            try:   # the reference's float64 arithmetic on the same workload (strict-parity mode, generic kernels)
                out["also_f64"] = measure_f64("c2")
            except Exception as ex:
                out["also_f64"] = {"error": repr(ex)[:300]}
            try:   # BASELINE configs[4] in the default run: the learned-dynamics line, GPU part only
                c5 = measure_c5(WORKLOADS["c5"], 200, 20, cpu=False)
                out["also_c5"] = {k: c5[k] for k in ("value", "unit", "ms_per_step", "dtype", "config", "roofline")}
            except Exception as ex:
                out["also_c5"] = {"error": repr(ex)[:300]}
        if not args.no_cpu_baseline and world == 1 and args.cost_mode == "sum" and w.get("env") not in ("door", "relocate", "fpp"):   # the C oracle's loop: sum, HalfCheetah / HumanoidStandup form
            out["cpu_baseline"] = cpu_baseline(args.workload, c_port_threads(w))
            out["cpu_baseline"]["host_logical_cpus"] = usable_cores()
            out["cpu_baselines"] = extra_cpu_baselines(args.workload, w, model, env)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
