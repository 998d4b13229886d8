"""The distillation teacher of the search - Light-Weight RefineNet on ResNet-152 (Nekrasov et al., BMVC
2018) - running on the nasseg HIP kernels.

Interface and module tree of src/kd/rf_lw/model_lw_v2.py: ``rf_lw152(pretrained, num_classes)`` ->
``ResNetLW``; attribute names, construction order (hence seeded initialisation) and ``state_dict`` keys /
shapes are the reference's, so its published checkpoint loads unchanged (``load_state_dict``).  The engine
only ever calls ``kd_net(image)`` under ``no_grad`` in eval mode (src/engine/trainer.py:17-74,
populate_task0) and keeps the bilinearly resized logits in the task0 cache; the forward here is that
inference path: every conv + BatchNorm (+ ReLU, + the block's skip connection) is ONE kernel (BatchNorm
folded into the conv's epilogue), max-pools 3x3 / 5x5 (the chained residual pooling blocks), the
decoder's ``align_corners=True`` up-sampling (nasseg_bilinear_ac_fwd), adds and ReLUs are nasseg kernels;
nothing reaches ATen.  Dropout is the identity in eval mode and is refused in training mode (the teacher
is never trained; there is no dropout kernel).  No checkpoint can be downloaded here: ``pretrained=True``
looks for the file the reference would have cached and fails loudly otherwise.
"""
import os

import torch
import torch.nn as nn

from .. import functional as F
from .._lib import NassegError
from ..nn.modules import BatchNorm2d, Conv2d, FusedSequential, MaxPool2d, ReLU, run_fused

num_classes = 21

# ResNet stages: (bottleneck width, stride of the first block); depths come with the variant
_RESNET_STAGES = ((64, 1), (128, 2), (256, 2), (512, 2))
_EXPANSION = 4
# The RefineNet decoder, deepest encoder map first: (encoder channels, width of the stage, channels handed to
# the next stage or None for the last one).  Attribute names follow the reference checkpoint
# (model_lw_v2.py:191-211): p_ims1d2_outl{g}_dimred, adapt_stage{g}_b2_joint_varout_dimred (g > 1),
# mflow_conv_g{g}_pool, mflow_conv_g{g}_b3_joint_varout_dimred (g < 4) - created in exactly this order, which
# is also the order a seeded initialisation visits them in.
_DECODER_STAGES = ((2048, 512, 256), (1024, 256, 256), (512, 256, 256), (256, 256, None))
_CRP_STAGES = 4


def _pointwise(cin, cout):
    return Conv2d(cin, cout, kernel_size=1, stride=1, padding=0, bias=False)


class CRPBlock(nn.Module):
    """Chained residual pooling (model_lw_v2.py:76-100): a running sum of n x [5x5 max-pool, stride 1 -> 1x1
    conv] applied to the block's input.  Keys ``{i}_outvar_dimred.weight``, i = 1..n."""

    def __init__(self, in_planes, out_planes, n_stages):
        super().__init__()
        self.n_stages = n_stages
        self.stride = 1
        widths = [in_planes] + [out_planes] * n_stages
        for i in range(n_stages):
            self.add_module("{}_outvar_dimred".format(i + 1), _pointwise(widths[i], widths[i + 1]))
        self.maxpool = MaxPool2d(kernel_size=5, stride=1, padding=2)

    def forward(self, x):
        acc, path = x, x
        for i in range(self.n_stages):
            path = self._modules["{}_outvar_dimred".format(i + 1)](self.maxpool(path))
            acc = F.add(path, acc)
        return acc


class Bottleneck(nn.Module):
    """1x1 -> 3x3 (stride) -> 1x1 (x4) with BatchNorms, skip connection, ReLU (model_lw_v2.py:131-169); the whole
    body is one fused conv chain with the skip as its residual."""
    expansion = _EXPANSION

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = BatchNorm2d(planes, momentum=0.95)
        self.conv2 = Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = BatchNorm2d(planes, momentum=0.95)
        self.conv3 = Conv2d(planes, planes * _EXPANSION, kernel_size=1, bias=False)
        self.bn3 = BatchNorm2d(planes * _EXPANSION, momentum=0.95)
        self.relu = ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        body = [self.conv1, self.bn1, self.relu, self.conv2, self.bn2, self.relu, self.conv3, self.bn3]
        return F.relu(run_fused(body, x, residual=skip))


class ResNetLW(nn.Module):
    """Encoder (ResNet) + Light-Weight RefineNet decoder, both generated from the tables above."""

    def __init__(self, block, layers, num_classes=21):
        super().__init__()
        self.inplanes = 64
        self.do = nn.Dropout(p=0.5)
        self.conv1 = Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.relu = ReLU(inplace=True)
        self.maxpool = MaxPool2d(kernel_size=3, stride=2, padding=1)
        for i, ((planes, stride), depth) in enumerate(zip(_RESNET_STAGES, layers)):
            self.add_module("layer{}".format(i + 1), self._make_layer(block, planes, depth, stride))
        for g, (enc, width, out) in enumerate(_DECODER_STAGES, start=1):
            self.add_module("p_ims1d2_outl{}_dimred".format(g), _pointwise(enc, width))
            if g > 1:
                self.add_module("adapt_stage{}_b2_joint_varout_dimred".format(g), _pointwise(width, width))
            self.add_module("mflow_conv_g{}_pool".format(g), nn.Sequential(CRPBlock(width, width, _CRP_STAGES)))
            if out is not None:
                self.add_module("mflow_conv_g{}_b3_joint_varout_dimred".format(g), _pointwise(width, out))
        self.clf_conv = Conv2d(_DECODER_STAGES[-1][1], num_classes, kernel_size=3, stride=1, padding=1, bias=True)

    def _make_layer(self, block, planes, blocks, stride=1):
        width = planes * block.expansion
        project = None
        if stride != 1 or self.inplanes != width:
            project = FusedSequential(Conv2d(self.inplanes, width, kernel_size=1, stride=stride, bias=False),
                                      BatchNorm2d(width))
        units = [block(self.inplanes, planes, stride, project)]
        units += [block(width, planes) for _ in range(blocks - 1)]
        self.inplanes = width
        return nn.Sequential(*units)

    def forward(self, x):
        if self.training:
            raise NassegError("the distillation teacher is inference-only here (no dropout kernel): call .eval()")
        x = self.maxpool(run_fused([self.conv1, self.bn1, self.relu], x))
        pyramid = []
        for i in range(len(_RESNET_STAGES)):
            x = self._modules["layer{}".format(i + 1)](x)
            pyramid.append(x)
        pyramid.reverse()  # deepest first; (dropout on the two deepest maps is the identity in eval mode)
        m = self._modules
        carried = None
        for g, feat in enumerate(pyramid, start=1):
            y = m["p_ims1d2_outl{}_dimred".format(g)](feat)
            if carried is not None:
                y = F.add(m["adapt_stage{}_b2_joint_varout_dimred".format(g)](y), carried)
            y = m["mflow_conv_g{}_pool".format(g)](F.relu(y))
            if g < len(pyramid):
                y = m["mflow_conv_g{}_b3_joint_varout_dimred".format(g)](y)
                carried = F.bilinear_resize(y, pyramid[g].size()[2:], align_corners=True)
        return self.clf_conv(y)


def _checkpoint_path(name):
    """Where the reference's loader caches its download (model_lw_v2.py:283-289 -> utils/helpers.maybe_download)."""
    torch_home = os.path.expanduser(os.getenv("TORCH_HOME", "~/.torch"))
    return os.path.join(os.getenv("TORCH_MODEL_ZOO", os.path.join(torch_home, "models")), name)


def rf_lw152(pretrained=False, num_classes=num_classes, **kwargs):
    """ResNet-152 Light-Weight RefineNet (model_lw_v2.py:280-297).  ``pretrained=True`` loads the reference's
    checkpoint from its cache location (entries whose key this model lacks are ignored, as there); there is no
    network here, so a missing file is an error."""
    model = ResNetLW(Bottleneck, [3, 8, 36, 3], num_classes=num_classes, **kwargs)
    if pretrained:
        path = _checkpoint_path("rf_lw152.pth.tar")
        if not os.path.exists(path):
            raise NassegError("rf_lw152(pretrained=True): {} not found and there is no network here; fetch the "
                              "reference's checkpoint to that path".format(path))
        known = model.state_dict()
        loaded = {k: v for k, v in torch.load(path, map_location="cpu").items() if k in known}
        model.load_state_dict(loaded, strict=False)
    return model
