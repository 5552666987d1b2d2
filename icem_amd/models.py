"""Forward-model side of the boundary.

``ForwardModel`` mirrors the reference's batched model contract
(icem/misc/base_types.py:62-118, icem/models/abstract_models.py:8-53): ``predict`` maps
``([N,o], states, [N,d]) -> ([N,o], states, [N,1])`` and ``predict_n_steps`` rolls a policy out
for ``horizon`` steps.  ``DeviceSyntheticModel`` is the built-in analytic model the HIP rollout
kernel evaluates on the GPU; its NumPy ``predict`` exists so the same object can be driven
through the reference-style interface (used by the host-model path and by tests).
"""
from __future__ import annotations

import math
from abc import ABC, abstractmethod

import numpy as np

MODEL_LINEAR, MODEL_TANH = 0, 1


class TrajectoryBatch:
    """What ``predict_n_steps`` returns here instead of N ``Rollout`` objects (the reference
    spends ~90 % of a step building those -- icem/misc/rolloutbuffer.py:16-39,155-172): flat
    arrays ``[N,h,.]``.  Indexing yields per-trajectory dict views so reference-style consumers
    (``r["observations"]``, ``buffer.as_array("actions")``) keep working."""

    fields = ("observations", "next_observations", "actions", "rewards")

    def __init__(self, **arrays):
        self._a = {k: np.asarray(v) for k, v in arrays.items()}
        n = {len(v) for v in self._a.values()}
        if len(n) > 1:
            raise TypeError("Turning rollout structure into numpy array failed. Rollouts of unequal length?")

    def __len__(self):
        return len(next(iter(self._a.values()))) if self._a else 0

    def __bool__(self):
        return len(self) > 0

    def as_array(self, key):
        return self._a[key]

    def __getitem__(self, item):
        if isinstance(item, str):
            return self._a[item].reshape((-1,) + self._a[item].shape[2:])
        if isinstance(item, (int, np.integer)):
            return {k: v[item] for k, v in self._a.items()}
        return TrajectoryBatch(**{k: v[item] for k, v in self._a.items()})

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    # -- the mutating half of RolloutBuffer's sequence interface (icem/misc/rolloutbuffer.py:112-123, 156-172, 277) ----------
    @property
    def is_empty(self):
        return len(self) == 0

    def extend(self, other):
        """Add trajectories: another ``TrajectoryBatch`` (its arrays are concatenated along N) or an iterable of
        per-trajectory dict views as ``self[i]`` hands them out.  An empty batch takes the other's fields; otherwise the
        fields must be the same and of the same horizon -- the reference's "Rollouts of unequal length?" error."""
        if not isinstance(other, TrajectoryBatch):
            rows = list(other)
            if not rows:
                return
            try:
                other = TrajectoryBatch(**{k: np.stack([np.asarray(r[k]) for r in rows]) for k in rows[0]})
            except (ValueError, KeyError) as e:
                raise TypeError(f"Concatenating rollouts failed with error {e}")
        if len(other) == 0:
            return
        if not self._a:
            self._a = {k: np.array(v) for k, v in other._a.items()}
            return
        if set(other._a) != set(self._a) or any(other._a[k].shape[1:] != v.shape[1:] for k, v in self._a.items()):
            raise TypeError("Turning rollout structure into numpy array failed. Rollouts of unequal length?")
        self._a = {k: np.concatenate([v, other._a[k]], axis=0) for k, v in self._a.items()}

    def append(self, item):
        self.extend([item])

    @property
    def flat(self):
        """All transitions of all trajectories, ``{field: [N * h, ...]}`` (``RolloutBuffer.flat``: the concatenation of the
        rollouts' transition records); per-trajectory scalars such as ``costs`` are left out."""
        return {k: v.reshape((-1,) + v.shape[2:]) for k, v in self._a.items() if v.ndim >= 2}


class ForwardModel(ABC):
    supports_stochastic = False

    def __init__(self, *, env=None):
        self.env = env

    def reset(self, observation):
        return None

    def got_actual_observation_and_env_state(self, *, observation, env_state=None, model_state=None):
        return None

    @abstractmethod
    def predict(self, *, observations, states, actions):
        """-> (next_observations [N,o], next_states, rewards [N,1])"""

    def rollout_generator(self, start_states, start_observations, horizon, policy, mode=None):
        states, obs = start_states, start_observations
        for _ in range(horizon):
            actions = policy.get_action(obs, state=states, mode=mode)
            next_obs, next_states, r = self.predict(observations=obs, states=states, actions=actions)
            yield obs, next_obs, actions, states, r
            states, obs = next_states, next_obs

    def rollout_field_names(self):
        return TrajectoryBatch.fields

    def predict_n_steps(self, *, start_observations, start_states, policy, horizon):
        if start_observations.ndim != 2:
            raise AttributeError("call predict_n_steps with a batch of states")
        if len(start_observations) != len(start_states):
            raise AttributeError("number of observations and states have to be the same")
        obs, nxt, act, states, rew = zip(*self.rollout_generator(start_states, start_observations, horizon, policy))
        tr = lambda x: np.asarray(x).transpose((1, 0, 2))
        return TrajectoryBatch(observations=tr(obs), next_observations=tr(nxt), actions=tr(act),
                               rewards=tr(rew)), states[-1]

    def train(self, buffer):
        pass

    def save(self, path):
        pass

    def load(self, path):
        pass


class DeviceSyntheticModel(ForwardModel):
    """``o' = act(o @ A + a @ B)``, ``A [o,o]``, ``B [d,o]``; ``kind`` linear or tanh; ``band >= 0``
    zeroes ``A`` outside ``|row-col| <= band``."""

    def __init__(self, A, B, kind: int = MODEL_LINEAR, band: int = -1, env=None):
        super().__init__(env=env)
        A = np.array(A, dtype=np.float64)
        B = np.array(B, dtype=np.float64)
        if band >= 0:
            r = np.arange(A.shape[0])
            A = np.where(np.abs(r[:, None] - r[None, :]) <= band, A, 0.0)
        self.A, self.B, self.kind, self.band = A, B, kind, band
        self.obs_dim, self.act_dim = A.shape[0], B.shape[0]

    @staticmethod
    def make(obs_dim: int, act_dim: int, kind: int = MODEL_LINEAR, band: int = -1, seed_a: int = 0,
             seed_b: int = 1, env=None) -> "DeviceSyntheticModel":
        """The synthetic dynamics of the benchmark configs: ``A = 0.95 I + 0.05 N(0,1)/sqrt(o)``
        (RandomState(seed_a)), ``B = 0.1 N(0,1)`` (RandomState(seed_b))."""
        A = 0.95 * np.eye(obs_dim) + 0.05 * np.random.RandomState(seed_a).randn(obs_dim, obs_dim) / math.sqrt(obs_dim)
        B = 0.1 * np.random.RandomState(seed_b).randn(act_dim, obs_dim)
        return DeviceSyntheticModel(A, B, kind, band, env)

    def predict(self, *, observations, states, actions):
        observations = np.asarray(observations, dtype=np.float64)
        actions = np.asarray(actions, dtype=np.float64)
        nxt = np.zeros_like(observations)
        for k in range(self.obs_dim):
            nxt = nxt + observations[..., k:k + 1] * self.A[k]
        for j in range(self.act_dim):
            nxt = nxt + actions[..., j:j + 1] * self.B[j]
        if self.kind == MODEL_TANH:
            nxt = np.tanh(nxt)
        # a model that carries its environment reports the environment's reward, like the reference's ground-truth
        # models do (env.step's reward = -cost_fn); without one there is no reward to report
        if self.env is not None and hasattr(self.env, "reward_fn"):
            rew = np.asarray(self.env.reward_fn(observations, actions, nxt), dtype=np.float64)[..., None]
        else:
            rew = np.zeros(observations.shape[:-1] + (1,))
        return nxt, None, rew


class TorchForwardModel(ForwardModel):
    """Adapter for a learned, device-resident dynamics model (the README's PlaNet column has no
    code in the reference, so the architecture is the caller's).  ``module(obs[N,o], act[N,d]) -> next_obs[N,o]``
    is any ``torch.nn.Module`` on the GPU (its GEMMs run on the matrix cores through hipBLASLt, bf16 if the
    module is bf16); ``cost(obs, act) -> [N]`` is a torch callable.  ``MpcICemHip`` keeps the whole CEM loop on
    the device with it; ``predict`` serves the reference-style NumPy interface."""

    def __init__(self, module, cost, obs_dim: int, act_dim: int, dtype=None, device="cuda:0", env=None, use_graph=True):
        super().__init__(env=env)
        import torch
        self.use_graph = use_graph   # replay the h-step chain of torch launches as a HIP graph (MpcController._costs_of)
        self.module = module.to(device)
        self.cost = cost
        self.obs_dim, self.act_dim = obs_dim, act_dim
        self.device = torch.device(device)
        self.dtype = dtype or next(module.parameters()).dtype

    def torch_step(self, obs, act):
        import torch
        with torch.no_grad():
            return self.module(obs, act)

    def torch_cost(self, obs, act):
        import torch
        with torch.no_grad():
            return self.cost(obs, act)

    def predict(self, *, observations, states, actions):
        import torch
        o = torch.as_tensor(np.asarray(observations), dtype=self.dtype, device=self.device)
        a = torch.as_tensor(np.asarray(actions), dtype=self.dtype, device=self.device)
        single = o.ndim == 1
        if single:
            o, a = o[None], a[None]
        nxt = self.torch_step(o, a).float().cpu().numpy().astype(np.float64)
        if single:
            nxt = nxt[0]
        return nxt, None, np.zeros(nxt.shape[:-1] + (1,))


def declared_rssm(act_dim: int = 6, det: int = 200, stoch: int = 30, hidden: int = 200, seed: int = 0, dtype=None,
                  device="cuda:0", env=None) -> TorchForwardModel:
    """The learned-dynamics stand-in declared for BASELINE config 5 (N=1024, h=12: "PlaNet learned dynamics" -- the
    reference ships no such model, README.md:21-29 only quotes its results): a recurrent state-space model in the
    PlaNet shape, rolled out on its prior mean.  Planner observation = ``[h (det) | z (stoch)]``;
    ``x = relu(W1 [z, a])``, ``h' = GRUCell(x, h)``, ``z' = W3 relu(W2 h')`` and a two-layer reward head on
    ``[h', z']``; cost of a step = minus the predicted reward of the state it starts from.  Random weights
    (``seed``); ``dtype=torch.bfloat16`` runs the GEMMs on the bf16 matrix cores through hipBLASLt."""
    import torch

    class RSSM(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.det, self.stoch = det, stoch
            self.inp = torch.nn.Linear(stoch + act_dim, hidden)
            self.gru = torch.nn.GRUCell(hidden, det)
            self.prior1 = torch.nn.Linear(det, hidden)
            self.prior2 = torch.nn.Linear(hidden, stoch)
            self.rew1 = torch.nn.Linear(det + stoch, hidden)
            self.rew2 = torch.nn.Linear(hidden, hidden)
            self.rew3 = torch.nn.Linear(hidden, 1)

        def forward(self, obs, act):
            h, z = obs[:, :self.det], obs[:, self.det:]
            x = torch.relu(self.inp(torch.cat([z, act], dim=-1)))
            h2 = self.gru(x, h)
            z2 = self.prior2(torch.relu(self.prior1(h2)))
            return torch.cat([h2, z2], dim=-1)

        def reward(self, obs):
            return self.rew3(torch.relu(self.rew2(torch.relu(self.rew1(obs))))).squeeze(-1)

    gen = torch.Generator().manual_seed(seed)
    net = RSSM()
    with torch.no_grad():
        for p in net.parameters():   # reproducible weights, independent of torch's global RNG state
            bound = 1.0 / math.sqrt(p.shape[-1]) if p.ndim > 1 else 0.05
            p.copy_((torch.rand(p.shape, generator=gen) * 2 - 1) * bound)
    if dtype is not None:
        net = net.to(dtype)
    return TorchForwardModel(net, lambda o, a: -net.reward(o), det + stoch, act_dim, dtype=dtype, device=device, env=env)


def pack_rssm(module):
    """The packed bf16 parameter buffer ``icem_rssm_rollout_cost`` reads (layout: icem_amd/csrc/icem_rssm.h) from a
    ``declared_rssm`` module: every weight padded (outputs to multiples of 16, contraction to multiples of 32), cut
    into 16 x 32 blocks stored as ``v_mfma_f32_16x16x32_bf16`` A operands ``[out block][k block][lane = 16*g + i][v]``
    = ``W[16*ob + i][32*kb + 8*g + v]``, each layer followed by its f32 bias."""
    import torch
    sd = {k: v.detach().float().cpu() for k, v in module.state_dict().items()}
    det, st = module.det, module.stoch
    assert (det, st) == (200, 30) and sd["inp.weight"].shape == (200, 36), "icem_rssm_rollout_cost is compiled for the declared sizes"

    def pad(w, rows, cols, col_map=None):
        out = torch.zeros(rows, cols)
        if col_map is None:
            out[:w.shape[0], :w.shape[1]] = w
        else:
            for (s0, s1, d0) in col_map:
                out[:w.shape[0], d0:d0 + (s1 - s0)] = w[:, s0:s1]
        return out

    def gates(w, cols):   # [3*200, k] -> [3*208, cols]: each gate padded separately
        return torch.cat([pad(w[i * det:(i + 1) * det], 208, cols) for i in range(3)], dim=0)

    def gate_bias(b):
        return torch.cat([pad(b[i * det:(i + 1) * det][None], 1, 208)[0] for i in range(3)])

    layers = [
        (pad(sd["inp.weight"], 208, 64, [(0, st, 0), (st, st + 6, 32)]), pad(sd["inp.bias"][None], 1, 208)[0]),
        (gates(sd["gru.weight_ih"], 224), gate_bias(sd["gru.bias_ih"])),
        (gates(sd["gru.weight_hh"], 224), gate_bias(sd["gru.bias_hh"])),
        (pad(sd["prior1.weight"], 208, 224), pad(sd["prior1.bias"][None], 1, 208)[0]),
        (pad(sd["prior2.weight"], 32, 224), pad(sd["prior2.bias"][None], 1, 32)[0]),
        (pad(sd["rew1.weight"], 208, 256, [(0, det, 0), (det, det + st, 224)]), pad(sd["rew1.bias"][None], 1, 208)[0]),
        (pad(sd["rew2.weight"], 208, 224), pad(sd["rew2.bias"][None], 1, 208)[0]),
        (pad(sd["rew3.weight"], 16, 224), pad(sd["rew3.bias"][None], 1, 16)[0]),
    ]
    chunks = []
    for w, b in layers:
        ob, kb = w.shape[0] // 16, w.shape[1] // 32
        blocks = w.reshape(ob, 16, kb, 4, 8).permute(0, 2, 3, 1, 4).contiguous()   # [ob, kb, g, i, v]
        chunks.append(blocks.to(torch.bfloat16).view(torch.int16).reshape(-1))
        chunks.append(b.contiguous().view(torch.int16).reshape(-1))                 # f32 bias as two 16-bit words each
    return torch.cat(chunks)


class DeviceRSSMModel(ForwardModel):
    """The declared RSSM with its whole h-step rollout + cost fused into one HIP launch on the bf16 matrix cores
    (``icem_rssm_rollout_cost``); ``MpcICemHip`` scores a population with it in a single kernel.  ``reference`` is the
    f32 torch module the weights come from (``predict`` serves the reference-style NumPy interface through it)."""

    def __init__(self, seed: int = 0, device="cuda:0", env=None):
        super().__init__(env=env)
        import torch
        from . import _lib as L
        self._tm = declared_rssm(seed=seed, device=device)
        self.reference = self._tm.module
        self.obs_dim, self.act_dim = 230, 6
        self.device = torch.device(device)
        self.params = pack_rssm(self.reference).to(self.device)
        self.lib = L.load_library()
        L.maybe_follow_environment()   # (tools only: icem_amd._lib.follow_environment)
        self._obs_host = self._obs_dev = None
        if self.params.numel() != self.lib.icem_rssm_param_elems():
            raise RuntimeError("packed RSSM parameters do not match the library's layout")

    def rollout_cost(self, obs, actions, cost_mode: int = 0):
        """costs ``[n]`` (f32 device tensor) of f32 device action sequences ``[n, h, 6]`` from observation ``obs [230]``."""
        import ctypes as C
        import torch
        from . import _lib as L
        actions = actions.to(torch.float32).contiguous()
        n, h, _ = actions.shape
        ob = np.asarray(obs, dtype=np.float32)
        if self._obs_host is None or not np.array_equal(ob, self._obs_host):   # the CEM iterations of a step share it
            self._obs_host, self._obs_dev = ob.copy(), torch.as_tensor(ob, device=self.device)
        o = self._obs_dev
        costs = torch.empty((n,), dtype=torch.float32, device=self.device)
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        L.check(self.lib.icem_rssm_rollout_cost(n, h, cost_mode, C.c_void_p(self.params.data_ptr()), C.c_void_p(o.data_ptr()),
                                                C.c_void_p(actions.data_ptr()), C.c_void_p(costs.data_ptr()), st))
        return costs

    def predict(self, *, observations, states, actions):
        return self._tm.predict(observations=observations, states=states, actions=actions)
