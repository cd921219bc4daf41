from .atsp import ATSPEnv, ATSPGenerator  # noqa: F401
from .rcvrp import RCVRPEnv, RCVRPGenerator  # noqa: F401
