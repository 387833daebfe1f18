"""Nested-class configuration objects (same contract as legged_gym/envs/base/base_config.py:3-25:
instantiating the outer class replaces every nested class attribute by an instance, recursively)."""
import inspect


class BaseConfig:
    def __init__(self):
        _instantiate_members(self)


def _instantiate_members(obj):
    for name in dir(obj):
        if name == "__class__":
            continue
        member = getattr(obj, name)
        if inspect.isclass(member):
            inst = member()
            setattr(obj, name, inst)
            _instantiate_members(inst)
