"""NAS inner loop: candidate training steps and validation (mirrors src/engine)."""
from .segmenter import PeerFailure, RankParallel, Segmenter  # noqa: F401
