"""Baselines of the reference that run on the MI355X engine (rrnco/baselines): the MatNet baseline (mixed-score attention
encoder, attention-model decoder, the baseline's own decoding rules)."""
from .matnet import MatNetDecoder, MatNetEncoder, MatNetPolicy

__all__ = ["MatNetEncoder", "MatNetDecoder", "MatNetPolicy"]
