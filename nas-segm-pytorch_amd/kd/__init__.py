"""Knowledge-distillation teacher (src/kd/rf_lw) on the nasseg kernels."""
from .rf_lw import ResNetLW, rf_lw152  # noqa: F401
