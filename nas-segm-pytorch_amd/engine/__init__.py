"""NAS inner loop: candidate training steps and validation (mirrors src/engine)."""
from .segmenter import RankParallel, Segmenter  # noqa: F401
