"""MI355X-native implementation of the NAS inner loop of DrSleep/nas-segm-pytorch.

Import as ``nas_segm_amd`` (the repo-root shim maps that name onto this
directory, whose on-disk name is not a valid Python identifier).
"""
__version__ = "0.1.0"

from . import functional  # noqa: F401
from ._lib import NassegError, lib  # noqa: F401


def install_dropin(kd=False, data=False):
    """Register this package under the module names the reference's own scripts
    import (``nn.layer_factory``, ``nn.micro_decoders``, ``nn.encoders``,
    ``rl.genotypes``, ``helpers.miou_utils``, ``engine.trainer``,
    ``engine.inference``) so that e.g. the reference's src/main_search.py and
    tests/test_inference.py resolve to the HIP implementation unchanged.
    kd=True also maps ``kd.rf_lw.model_lw_v2`` (the distillation teacher,
    src/main_search.py:456) onto ``kd/rf_lw.py``; data=True maps ``data.loaders``
    / ``data.datasets`` (src/main_search.py:30) onto ``data/`` - no OpenCV needed,
    its resizes are restatements whose parity with cv2 is NOT pinned.
    See INTEGRATION.md."""
    import sys

    from . import engine, helpers, nn, rl
    from .engine import inference, trainer
    from .helpers import miou_utils
    from .nn import encoders, layer_factory, micro_decoders
    from .rl import genotypes

    table = {
        "nn": nn, "nn.layer_factory": layer_factory, "nn.micro_decoders": micro_decoders,
        "nn.encoders": encoders, "rl.genotypes": genotypes, "helpers.miou_utils": miou_utils,
        "engine.trainer": trainer, "engine.inference": inference,
    }
    if kd:
        from . import kd as kd_pkg
        from .kd import rf_lw

        table.update({"kd": kd_pkg, "kd.rf_lw": rf_lw, "kd.rf_lw.model_lw_v2": rf_lw})
    if data:
        from . import data as data_pkg
        from .data import datasets, loaders

        table.update({"data": data_pkg, "data.datasets": datasets, "data.loaders": loaders})
    for name, mod in table.items():
        sys.modules[name] = mod
    # `rl`, `helpers`, `engine` keep the reference's other submodules importable:
    # only the hot-path submodules above are replaced.
    return sorted(table)
