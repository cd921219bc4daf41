"""Baselines of the reference that run on the MI355X engine (rrnco/baselines): the MatNet mixed-score attention encoder."""
from .matnet import MatNetEncoder

__all__ = ["MatNetEncoder"]
