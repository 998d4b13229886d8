"""MobileNetV2 encoder truncated at the deepest returned stage (mirrors
src/nn/encoders.py:15-83): ``forward(x) -> [feature map per return layer]`` and
an ``out_sizes`` list (channels per returned map) consumed by the decoders."""
import torch
import torch.nn as nn

from .layer_factory import InvertedResidual, conv_bn_relu6

__all__ = ["mbv2"]

model_paths = {"mbv2_voc": "./data/weights/mbv2_voc_rflw.ckpt"}


class MobileNetV2(nn.Module):
    # (expansion t, output channels c, repeats n, first stride s) per stage
    mobilenet_config = [
        [1, 16, 1, 1],
        [6, 24, 2, 2],
        [6, 32, 3, 2],
        [6, 64, 4, 2],
        [6, 96, 3, 1],
        [6, 160, 3, 2],
        [6, 320, 1, 1],
    ]
    in_planes = 32
    num_layers = len(mobilenet_config)

    def __init__(self, width_mult=1.0, return_layers=[1, 2, 4, 6]):
        super(MobileNetV2, self).__init__()
        self.return_layers = return_layers
        self.max_layer = max(return_layers)
        self.out_sizes = [self.mobilenet_config[i][1] for i in return_layers]
        width = int(self.in_planes * width_mult)
        self.layer1 = conv_bn_relu6(3, width, 2)
        for stage, (t, c, n, s) in enumerate(self.mobilenet_config[: self.max_layer + 1]):
            out_width = int(c * width_mult)
            blocks = []
            for i in range(n):
                blocks.append(InvertedResidual(width, out_width, s if i == 0 else 1, t))
                width = out_width
            setattr(self, "layer{}".format(stage + 2), nn.Sequential(*blocks))

    def forward(self, x):
        x = self.layer1(x)
        stage_outs = []
        for stage in range(self.max_layer + 1):
            x = getattr(self, "layer{}".format(stage + 2))(x)
            stage_outs.append(x)
        return [stage_outs[i] for i in self.return_layers]


def mbv2(pretrained=False, **kwargs):
    """MobileNetV2 encoder; ``pretrained`` names a local checkpoint key as in the reference."""
    model = MobileNetV2(**kwargs)
    if pretrained:
        model.load_state_dict(torch.load(model_paths["mbv2_{}".format(str(pretrained))]),
                              strict=False)
    return model


def create_encoder(pretrained="voc", ctrl_version="cvpr", **kwargs):
    return_layers = [1, 2, 4, 6] if ctrl_version == "cvpr" else [1, 2]
    return mbv2(pretrained=pretrained, return_layers=return_layers, **kwargs)
