"""icem_amd -- MI355X-native iCEM inner planning loop.

The hot path of martius-lab/iCEM's ``MpcICem.get_action`` (colored-noise
sampling + clip, batched rollout, per-trajectory cost, sorted top-k, mean/std
refit) as hand-written HIP for gfx950 behind a C ABI (``include/icem_hip.h``,
``libicem_hip.so``), with a host-side mirror of the reference's controller /
forward-model interface.  There is no CPU fallback: every operator raises if
the HIP library is missing.
"""
from ._lib import IcemError, lib_path, load_library  # noqa: F401
from .planner import IcemConfig, IcemPlanner  # noqa: F401
from .envs import (CostSpec, CostTerm, SyntheticEnv, door_env, relocate_env, halfcheetah_env, humanoid_standup_env, ant_env, hopper_env, humanoid_env,  # noqa: F401
                   reacher_env, fetch_pick_and_place_env, fetch_reach_env)
from .models import DeviceSyntheticModel, DeviceRSSMModel, TorchForwardModel, declared_rssm, pack_rssm  # noqa: F401
from .controllers import (MpcICemHip, MpcCemStdHip, MpcRandomHip, controller_from_string, ControllerFactory)  # noqa: F401

__version__ = "0.1.0"
