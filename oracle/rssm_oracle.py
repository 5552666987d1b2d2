"""NumPy restatement of the declared recurrent state-space model (icem_amd/models.py::declared_rssm) -- TEST
INFRASTRUCTURE ONLY, like everything under oracle/: the checker of tests/ and the timed CPU baseline of
``bench.py --workload c5``.  The reference ships no learned-dynamics model (README.md:21-29 quotes PlaNet results
only), so this restates the build's own declared architecture: ``x = relu(W1 [z, a])``, ``h' = GRUCell(x, h)`` (torch
gate order r, u, n), ``z' = W5 relu(W4 h')``, reward head ``W8 relu(W7 relu(W6 [h, z]))``; cost of a step = minus the
reward of the state it starts from."""
import numpy as np


def params_from_state_dict(sd) -> dict:
    return {k: np.asarray(v.detach().cpu().double().numpy() if hasattr(v, "detach") else v, dtype=np.float64) for k, v in sd.items()}


def step(P: dict, obs: np.ndarray, act: np.ndarray, det: int = 200) -> np.ndarray:
    h, z = obs[:, :det], obs[:, det:]
    x = np.maximum(np.concatenate([z, act], -1) @ P["inp.weight"].T + P["inp.bias"], 0)
    gi = x @ P["gru.weight_ih"].T + P["gru.bias_ih"]
    gh = h @ P["gru.weight_hh"].T + P["gru.bias_hh"]
    r = 1.0 / (1.0 + np.exp(-(gi[:, :det] + gh[:, :det])))
    u = 1.0 / (1.0 + np.exp(-(gi[:, det:2 * det] + gh[:, det:2 * det])))
    n = np.tanh(gi[:, 2 * det:] + r * gh[:, 2 * det:])
    h2 = (1 - u) * n + u * h
    z2 = np.maximum(h2 @ P["prior1.weight"].T + P["prior1.bias"], 0) @ P["prior2.weight"].T + P["prior2.bias"]
    return np.concatenate([h2, z2], -1)


def reward(P: dict, obs: np.ndarray) -> np.ndarray:
    a = np.maximum(obs @ P["rew1.weight"].T + P["rew1.bias"], 0)
    a = np.maximum(a @ P["rew2.weight"].T + P["rew2.bias"], 0)
    return (a @ P["rew3.weight"].T + P["rew3.bias"])[:, 0]


def rollout_costs(P: dict, obs0: np.ndarray, actions: np.ndarray, mode: str = "sum") -> np.ndarray:
    ob = np.broadcast_to(np.asarray(obs0, dtype=np.float64), (actions.shape[0], len(obs0))).copy()
    steps = []
    for t in range(actions.shape[1]):
        steps.append(-reward(P, ob))
        ob = step(P, ob, actions[:, t])
    s = np.stack(steps, 1)
    return {"sum": s.sum(1), "best": s.min(1), "final": s[:, -1]}[mode]
