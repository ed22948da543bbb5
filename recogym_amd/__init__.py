"""recogym_amd — MI355X-native vectorised reco-gym-v1 step loop.

Same public names as the reference's `recogym` package for the hot path:

    import recogym_amd as recogym
    env = recogym.make('reco-gym-v1'); env.init_gym({**recogym.env_1_args, 'random_seed': 42})
    logs = env.generate_logs(1000)                       # pandas DataFrame, reference columns
    recogym.test_agent(env, agent, 1000, 1000); recogym.verify_agents(env, 1000, {...})
"""
from .envs import Configuration, Context, DefaultContext, Observation, Session, OrganicSessions

_REGISTRY = {}


def register(id, entry_point, **kwargs):
    _REGISTRY[id] = entry_point


def make(env_id):
    """gym.make for the ids this package registers (reference: recogym/__init__.py:37-45)."""
    import importlib
    module, cls = _REGISTRY[env_id].split(':')
    return getattr(importlib.import_module(module), cls)()


register(id='reco-gym-v0', entry_point='recogym_amd.envs.reco_env_v0:RecoEnv0')      # reference: recogym/__init__.py:37-40
register(id='reco-gym-v1', entry_point='recogym_amd.envs.reco_env_v1:RecoEnv1')


def register_with_gym(env_id='reco-gym-v1', force=False):
    """Where OpenAI `gym` is importable, make `gym.make('reco-gym-v1')` resolve to this package's environment the
    way the reference registers its own (recogym/__init__.py:37-45).  Called once on import (like the reference's
    package does) when gym is installed and the id is still free; an id that is already registered — e.g. by the
    reference package imported earlier — is left alone unless force=True.  `gym.make` on a gym that wraps environments
    in checkers expects a gym.Env subclass: RecoEnv1 mirrors the reference's duck-typed surface (reset / step /
    step_offline / generate_logs), use recogym_amd.make for it.  Returns True when the id now points here."""
    try:
        from gym.envs.registration import register as gym_register
        import gym.envs.registration as reg
    except Exception:
        return False
    registry = getattr(reg, 'registry', None)
    specs = getattr(registry, 'env_specs', registry)
    try:
        known = specs is not None and env_id in specs
    except TypeError:
        known = False
    if known:
        if not force:
            return False
        try:
            del specs[env_id]
        except Exception:
            return False
    gym_register(id=env_id, entry_point=_REGISTRY[env_id])
    return True


def _auto_register():
    # `import recogym_amd as recogym; gym.make('reco-gym-v1')` — the reference's usage — works when gym is there
    import importlib.util
    try:
        if importlib.util.find_spec('gym') is not None:
            register_with_gym('reco-gym-v0')
            register_with_gym()
    except Exception:
        pass


_auto_register()


def __getattr__(name):
    # torch-dependent pieces load lazily so that `import recogym_amd` works everywhere
    if name in ('env_1_args', 'env_args', 'RecoEnv1'):
        from .envs import reco_env_v1
        return getattr(reco_env_v1, name)
    if name in ('env_0_args', 'RecoEnv0'):
        from .envs import reco_env_v0
        return getattr(reco_env_v0, name)
    if name == 'test_agent':
        from .bench_agents import test_agent
        return test_agent
    if name == 'verify_agents':
        from .evaluate_agent import verify_agents
        return verify_agents
    if name in ('Agent', 'RandomAgent', 'random_args', 'OrganicUserEventCounterAgent',
                'organic_user_count_args', 'LastViewTableAgent'):
        from . import agents
        return getattr(agents, name)
    raise AttributeError(name)
