"""MobileNetV2 feature extractor, cut off after the deepest stage a decoder asks for.

API of src/nn/encoders.py:15-83: ``mbv2(pretrained, return_layers=...)`` /
``create_encoder(pretrained, ctrl_version)`` build a module whose ``forward(image)`` returns one
feature map per entry of ``return_layers`` (indices into the seven bottleneck stages) and whose
``out_sizes`` lists the channel counts of those maps - what MicroDecoder / TemplateDecoder take
as ``inp_sizes``.  Module names (``layer1`` = stem, ``layer2`` ... ``layer8`` = stages) and hence
``state_dict`` keys are the reference's, so its checkpoints load unchanged.  Every conv / BN /
ReLU6 below runs as nasseg HIP kernels through ``layer_factory``'s fused sequences.
"""
from collections import namedtuple

import torch
import torch.nn as nn

from .layer_factory import InvertedResidual, conv_bn_relu6
from .modules import run_fused

__all__ = ["mbv2"]

# local checkpoint the reference ships for VOC (never downloaded here)
model_paths = {"mbv2_voc": "./data/weights/mbv2_voc_rflw.ckpt"}

Stage = namedtuple("Stage", "expansion channels blocks stride")
# the seven bottleneck stages of MobileNetV2 (Sandler et al. 2018, table 2); strides 1,2,2,2,1,2,1
# put stages 1 / 2 / 4 / 6 at 1/4, 1/8, 1/16 and 1/32 of the input resolution
_STAGES = (
    Stage(1, 16, 1, 1),
    Stage(6, 24, 2, 2),
    Stage(6, 32, 3, 2),
    Stage(6, 64, 4, 2),
    Stage(6, 96, 3, 1),
    Stage(6, 160, 3, 2),
    Stage(6, 320, 1, 1),
)
_STEM_CHANNELS = 32


def _scaled(channels, width_mult):
    return int(channels * width_mult)


class MobileNetV2(nn.Module):
    # kept under the reference's attribute names for code that introspects the class
    mobilenet_config = [list(st) for st in _STAGES]
    in_planes = _STEM_CHANNELS
    num_layers = len(_STAGES)

    def __init__(self, width_mult=1.0, return_layers=[1, 2, 4, 6]):
        super(MobileNetV2, self).__init__()
        self.return_layers = return_layers
        self.max_layer = max(return_layers)  # stages beyond it are never built
        self.out_sizes = [_STAGES[idx].channels for idx in return_layers]
        c_prev = _scaled(_STEM_CHANNELS, width_mult)
        self.layer1 = conv_bn_relu6(3, c_prev, 2)
        for idx in range(self.max_layer + 1):
            st = _STAGES[idx]
            c_out = _scaled(st.channels, width_mult)
            units = [InvertedResidual(c_prev if b == 0 else c_out, c_out, st.stride if b == 0 else 1,
                                      st.expansion) for b in range(st.blocks)]
            self.add_module(self._stage_name(idx), nn.Sequential(*units))
            c_prev = c_out

    @staticmethod
    def _stage_name(idx):
        return "layer{}".format(idx + 2)

    # Consecutive units (the stem, then every InvertedResidual block) whose boundary tensor nobody but
    # the next unit reads run as ONE fused sequence: the normalised activation between them is applied
    # as the next conv loads and never written (the stem's 32 channels and stage 1's 16 at half
    # resolution are the two largest tensors of the network).  A boundary is kept when the next block
    # adds its input back (skip connection) or when it is one of the returned feature maps.
    merge_units = True

    def forward(self, x):
        taps = {}
        run = [self.layer1]  # units of the sequence being collected; flushed at every kept boundary

        def flush(x):
            if len(run) == 1 and isinstance(run[0], InvertedResidual):
                x = run[0](x)
            else:
                mods = []
                for unit in run:
                    mods.append(unit.conv if isinstance(unit, InvertedResidual) else unit)
                x = run_fused(mods, x)
            del run[:]
            return x

        for idx in range(self.max_layer + 1):
            stage = getattr(self, self._stage_name(idx))
            units = list(stage)
            for b, unit in enumerate(units):
                # (a block with a skip connection ends its sequence too: its sum is a tensor of its own)
                fusable = (self.merge_units and isinstance(unit, InvertedResidual) and not unit.use_res_connect
                           and bool(run) and not getattr(run[-1], "use_res_connect", False))
                if not fusable and run:
                    x = flush(x)
                run.append(unit)
            if idx in self.return_layers:
                x = flush(x)
                taps[idx] = x
        if run:
            x = flush(x)
        return [taps[idx] for idx in self.return_layers]


def mbv2(pretrained=False, **kwargs):
    """MobileNetV2 encoder; a truthy ``pretrained`` names the local checkpoint ``mbv2_<name>``."""
    net = MobileNetV2(**kwargs)
    if pretrained:
        state = torch.load(model_paths["mbv2_{}".format(str(pretrained))])
        net.load_state_dict(state, strict=False)
    return net


def create_encoder(pretrained="voc", ctrl_version="cvpr", **kwargs):
    """Encoder for a controller family: four taps for the CVPR cells, two for the WACV templates."""
    taps = [1, 2, 4, 6] if ctrl_version == "cvpr" else [1, 2]
    return mbv2(pretrained=pretrained, return_layers=taps, **kwargs)
