"""Environment registry with the reference factory's names (helpers/factory.py:100-202)."""
from .. import specs


def environment_names():
  return specs.environment_names()


def get_environment_obj(name, *args, **kwargs):
  """Name -> a batched environment (num_envs defaults to 1).  Unknown name: NotImplementedError
  (helpers/factory.py:201-202)."""
  from ..environments import BatchedSafetyEnvironment
  if name not in specs.environment_names():
    raise NotImplementedError("The requested environment is not available: %s" % name)
  return BatchedSafetyEnvironment(name, *args, **kwargs)
