"""rrnco_amd — MI355X-native construction-rollout engine behind the rrnco.envs / rrnco.models API."""
from .tensordict_lite import TensorDict  # noqa: F401

__all__ = ["TensorDict", "envs", "models", "ops"]
