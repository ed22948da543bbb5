"""Minimal stand-in for the `gym` package (not installed here) so that the UNMODIFIED reference
under /root/reference can be imported by the fixture generator.
Test infrastructure only."""
import importlib

from . import spaces
from .envs import registration


class Env:
    def __init__(self):
        pass


def make(env_id):
    entry = registration.registry[env_id]
    module, cls = entry.split(':')
    return getattr(importlib.import_module(module), cls)()
