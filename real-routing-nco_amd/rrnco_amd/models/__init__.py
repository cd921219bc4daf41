from .decoder import RRNetDecoder  # noqa: F401
from .decoding import get_decoding_strategy  # noqa: F401
from .encoder import RRNetEncoder  # noqa: F401
from .policy import RRNetPolicy  # noqa: F401
