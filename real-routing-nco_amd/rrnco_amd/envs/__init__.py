from .atsp import ATSPEnv, ATSPGenerator  # noqa: F401
