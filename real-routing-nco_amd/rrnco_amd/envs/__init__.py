from .atsp import ATSPEnv, ATSPGenerator  # noqa: F401
from .rcvrp import RCVRPEnv, RCVRPGenerator  # noqa: F401
from .rmtvrp import RMTVRPEnv, RMTVRPGenerator  # noqa: F401
from .sampler import RealWorldSampler  # noqa: F401
