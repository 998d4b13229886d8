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


def conv3x3(in_planes, out_planes, stride=1, bias=False):
    return Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=bias)


def conv1x1(in_planes, out_planes, stride=1, bias=False):
    return Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, padding=0, bias=bias)


class CRPBlock(nn.Module):
    """Chained residual pooling (model_lw_v2.py:76-100): n x [5x5 max-pool (stride 1) -> 1x1 conv], each
    stage's output added to the running sum."""

    def __init__(self, in_planes, out_planes, n_stages):
        super(CRPBlock, self).__init__()
        for i in range(n_stages):
            setattr(self, "{}_{}".format(i + 1, "outvar_dimred"),
                    conv1x1(in_planes if (i == 0) else out_planes, out_planes, stride=1, bias=False))
        self.stride = 1
        self.n_stages = n_stages
        self.maxpool = MaxPool2d(kernel_size=5, stride=1, padding=2)

    def forward(self, x):
        top = x
        for i in range(self.n_stages):
            top = self.maxpool(top)
            top = getattr(self, "{}_{}".format(i + 1, "outvar_dimred"))(top)
            x = F.add(top, x)
        return x


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(BasicBlock, self).__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = BatchNorm2d(planes, momentum=0.95)
        self.relu = ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = BatchNorm2d(planes, momentum=0.95)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        residual = x if self.downsample is None else self.downsample(x)
        # conv1 -> bn1 -> relu -> conv2 -> bn2 (+ residual) as one fused sequence, then the ReLU
        out = run_fused([self.conv1, self.bn1, self.relu, self.conv2, self.bn2], x, residual=residual)
        return F.relu(out)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(Bottleneck, self).__init__()
        self.conv1 = Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = BatchNorm2d(planes, momentum=0.95)
        self.conv2 = Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = BatchNorm2d(planes, momentum=0.95)
        self.conv3 = Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = BatchNorm2d(planes * 4, momentum=0.95)
        self.relu = ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        residual = x if self.downsample is None else self.downsample(x)
        out = run_fused([self.conv1, self.bn1, self.relu, self.conv2, self.bn2, self.relu, self.conv3, self.bn3],
                        x, residual=residual)
        return F.relu(out)


class ResNetLW(nn.Module):
    def __init__(self, block, layers, num_classes=21):
        self.inplanes = 64
        super(ResNetLW, self).__init__()
        self.do = nn.Dropout(p=0.5)
        self.conv1 = Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.relu = ReLU(inplace=True)
        self.maxpool = MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.p_ims1d2_outl1_dimred = conv1x1(2048, 512, bias=False)
        self.mflow_conv_g1_pool = self._make_crp(512, 512, 4)
        self.mflow_conv_g1_b3_joint_varout_dimred = conv1x1(512, 256, bias=False)
        self.p_ims1d2_outl2_dimred = conv1x1(1024, 256, bias=False)
        self.adapt_stage2_b2_joint_varout_dimred = conv1x1(256, 256, bias=False)
        self.mflow_conv_g2_pool = self._make_crp(256, 256, 4)
        self.mflow_conv_g2_b3_joint_varout_dimred = conv1x1(256, 256, bias=False)

        self.p_ims1d2_outl3_dimred = conv1x1(512, 256, bias=False)
        self.adapt_stage3_b2_joint_varout_dimred = conv1x1(256, 256, bias=False)
        self.mflow_conv_g3_pool = self._make_crp(256, 256, 4)
        self.mflow_conv_g3_b3_joint_varout_dimred = conv1x1(256, 256, bias=False)

        self.p_ims1d2_outl4_dimred = conv1x1(256, 256, bias=False)
        self.adapt_stage4_b2_joint_varout_dimred = conv1x1(256, 256, bias=False)
        self.mflow_conv_g4_pool = self._make_crp(256, 256, 4)

        self.clf_conv = Conv2d(256, num_classes, kernel_size=3, stride=1, padding=1, bias=True)

    def _make_crp(self, in_planes, out_planes, stages):
        return nn.Sequential(CRPBlock(in_planes, out_planes, stages))

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = FusedSequential(
                Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                BatchNorm2d(planes * block.expansion),
            )
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def _dropout(self, x):
        if self.training:
            raise NassegError("the distillation teacher is inference-only here (no dropout kernel): call .eval()")
        return x

    def forward(self, x):
        x = run_fused([self.conv1, self.bn1, self.relu], x)
        x = self.maxpool(x)

        l1 = self.layer1(x)
        l2 = self.layer2(l1)
        l3 = self.layer3(l2)
        l4 = self.layer4(l3)

        l4 = self._dropout(l4)
        l3 = self._dropout(l3)

        x4 = F.relu(self.p_ims1d2_outl1_dimred(l4))
        x4 = self.mflow_conv_g1_pool(x4)
        x4 = self.mflow_conv_g1_b3_joint_varout_dimred(x4)
        x4 = F.bilinear_resize(x4, l3.size()[2:], align_corners=True)

        x3 = self.p_ims1d2_outl2_dimred(l3)
        x3 = self.adapt_stage2_b2_joint_varout_dimred(x3)
        x3 = F.relu(F.add(x3, x4))
        x3 = self.mflow_conv_g2_pool(x3)
        x3 = self.mflow_conv_g2_b3_joint_varout_dimred(x3)
        x3 = F.bilinear_resize(x3, l2.size()[2:], align_corners=True)

        x2 = self.p_ims1d2_outl3_dimred(l2)
        x2 = self.adapt_stage3_b2_joint_varout_dimred(x2)
        x2 = F.relu(F.add(x2, x3))
        x2 = self.mflow_conv_g3_pool(x2)
        x2 = self.mflow_conv_g3_b3_joint_varout_dimred(x2)
        x2 = F.bilinear_resize(x2, l1.size()[2:], align_corners=True)

        x1 = self.p_ims1d2_outl4_dimred(l1)
        x1 = self.adapt_stage4_b2_joint_varout_dimred(x1)
        x1 = F.relu(F.add(x1, x2))
        x1 = self.mflow_conv_g4_pool(x1)

        return self.clf_conv(x1)


def rf_lw152(pretrained=False, num_classes=num_classes, **kwargs):
    """ResNet-152 Light-Weight RefineNet (model_lw_v2.py:280-297)."""
    model = ResNetLW(Bottleneck, [3, 8, 36, 3], num_classes=num_classes, **kwargs)
    if pretrained:
        torch_home = os.path.expanduser(os.getenv("TORCH_HOME", "~/.torch"))
        model_dir = os.getenv("TORCH_MODEL_ZOO", os.path.join(torch_home, "models"))
        cached = os.path.join(model_dir, "rf_lw152.pth.tar")  # (where the reference caches its download)
        if not os.path.exists(cached):
            raise NassegError("rf_lw152(pretrained=True): {} not found and there is no network here; fetch the "
                              "reference's checkpoint to that path".format(cached))
        pretrained_dict = torch.load(cached, map_location="cpu")
        model_dict = model.state_dict()
        model_dict.update({k: v for k, v in pretrained_dict.items() if k in model_dict})
        model.load_state_dict(model_dict)
    return model
