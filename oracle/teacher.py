"""ORACLE - TEST INFRASTRUCTURE ONLY.  Functional restatement (PyTorch-CPU fp32) of the distillation
teacher's forward, Light-Weight RefineNet on ResNet-152 - src/kd/rf_lw/model_lw_v2.py - keyed by the
reference's ``state_dict`` names, eval mode (BatchNorm on its running statistics, dropout the identity):
the only way the engine uses it (``kd_net(image)`` under no_grad, src/engine/trainer.py:58-60).

Pinned by tests/golden/teacher.npz: logits and intermediate statistics of the imported reference with
seeded random weights (tests/golden/make_golden.py:gen_teacher; tests/test_oracle_golden.py)."""
import torch
import torch.nn.functional as F

LAYERS = (3, 8, 36, 3)  # rf_lw152: model_lw_v2.py:286
EXPANSION = 4


def _bn(sd, prefix, x):
    # (every BatchNorm2d of the model keeps torch's default eps; momentum only matters in training)
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                        sd[prefix + ".bias"], False, 0.0, 1e-5)


def _bottleneck(sd, prefix, x, stride):
    # model_lw_v2.py:139-180
    out = F.relu(_bn(sd, prefix + ".bn1", F.conv2d(x, sd[prefix + ".conv1.weight"])))
    out = F.relu(_bn(sd, prefix + ".bn2", F.conv2d(out, sd[prefix + ".conv2.weight"], None, stride, 1)))
    out = _bn(sd, prefix + ".bn3", F.conv2d(out, sd[prefix + ".conv3.weight"]))
    residual = x
    if prefix + ".downsample.0.weight" in sd:
        residual = _bn(sd, prefix + ".downsample.1", F.conv2d(x, sd[prefix + ".downsample.0.weight"], None, stride))
    return F.relu(out + residual)


def _layer(sd, name, x, blocks, stride):
    for i in range(blocks):
        x = _bottleneck(sd, "{}.{}".format(name, i), x, stride if i == 0 else 1)
    return x


def _crp(sd, prefix, x, stages=4):
    # model_lw_v2.py:76-100: 5x5 max-pool (stride 1, pad 2) -> 1x1 conv, summed into x, four times
    top = x
    for i in range(stages):
        top = F.max_pool2d(top, 5, 1, 2)
        top = F.conv2d(top, sd["{}.0.{}_outvar_dimred.weight".format(prefix, i + 1)])
        x = top + x
    return x


def _up(x, like):
    return F.interpolate(x, size=like.shape[2:], mode="bilinear", align_corners=True)


def rf_lw152(sd, x, taps=None):
    """logits (B, n_cls, H/4, W/4) of the teacher; ``taps``: a dict that receives l1..l4 (tests)"""
    x = F.relu(_bn(sd, "bn1", F.conv2d(x, sd["conv1.weight"], None, 2, 3)))
    x = F.max_pool2d(x, 3, 2, 1)
    l1 = _layer(sd, "layer1", x, LAYERS[0], 1)
    l2 = _layer(sd, "layer2", l1, LAYERS[1], 2)
    l3 = _layer(sd, "layer3", l2, LAYERS[2], 2)
    l4 = _layer(sd, "layer4", l3, LAYERS[3], 2)
    if taps is not None:
        taps.update(l1=l1, l2=l2, l3=l3, l4=l4)
    x4 = F.relu(F.conv2d(l4, sd["p_ims1d2_outl1_dimred.weight"]))
    x4 = _crp(sd, "mflow_conv_g1_pool", x4)
    x4 = _up(F.conv2d(x4, sd["mflow_conv_g1_b3_joint_varout_dimred.weight"]), l3)

    x3 = F.conv2d(F.conv2d(l3, sd["p_ims1d2_outl2_dimred.weight"]), sd["adapt_stage2_b2_joint_varout_dimred.weight"])
    x3 = _crp(sd, "mflow_conv_g2_pool", F.relu(x3 + x4))
    x3 = _up(F.conv2d(x3, sd["mflow_conv_g2_b3_joint_varout_dimred.weight"]), l2)

    x2 = F.conv2d(F.conv2d(l2, sd["p_ims1d2_outl3_dimred.weight"]), sd["adapt_stage3_b2_joint_varout_dimred.weight"])
    x2 = _crp(sd, "mflow_conv_g3_pool", F.relu(x2 + x3))
    x2 = _up(F.conv2d(x2, sd["mflow_conv_g3_b3_joint_varout_dimred.weight"]), l1)

    x1 = F.conv2d(F.conv2d(l1, sd["p_ims1d2_outl4_dimred.weight"]), sd["adapt_stage4_b2_joint_varout_dimred.weight"])
    x1 = _crp(sd, "mflow_conv_g4_pool", F.relu(x1 + x2))
    return F.conv2d(x1, sd["clf_conv.weight"], sd["clf_conv.bias"], 1, 1)
