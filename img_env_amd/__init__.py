"""img_env_amd -- MI355X-native batched step() path of DRL-Navigation/img_env.

    from img_env_amd import make_env, read_yaml
    env = make_env(read_yaml("cfg.yaml")); state = env.reset(); state, r, d, info = env.step(actions)

The simulation runs in ``csrc/libimgenv_hip.so`` (hand-written HIP for gfx950) behind the C ABI of
``include/imgenv.h``; there is no CPU fallback.
"""
from .config import params_from_cfg, read_yaml  # noqa: F401


def __getattr__(name):
    # envs / world import torch lazily so that `import img_env_amd` stays cheap
    if name in ("make_env", "ImageEnv", "ImageState", "ContinuousAction", "DiscreteActions", "wrapper_dict"):
        from . import envs
        return getattr(envs, name)
    if name == "VecImageEnv":
        from .vec_env import VecImageEnv
        return VecImageEnv
    if name == "World":
        from .world import World
        return World
    raise AttributeError(name)
