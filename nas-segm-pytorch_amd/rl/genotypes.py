"""Index -> op-registry-key tables of the search spaces.

Decoder configs produced by the reference controller (src/rl/micro_controllers.py:
148-265,442-571) are lists of *indices* into these tables, so their order is part
of the contract with src/rl/genotypes.py:8-35 and must not change.
"""
from collections import namedtuple

Genotype = namedtuple("Genotype", "encoder decoder")

# CVPR'19 search space (MicroDecoder cells): 11 ops
OP_NAMES = (
    "conv1x1 conv3x3 sep_conv_3x3 sep_conv_5x5 global_average_pool conv3x3_dil3 "
    "conv3x3_dil12 sep_conv_3x3_dil3 sep_conv_5x5_dil6 skip_connect none"
).split()

# WACV'20 search space (TemplateDecoder templates): 6 ops
OP_NAMES_WACV = (
    "sep_conv_3x3 sep_conv_5x5 global_average_pool max_pool_3x3 sep_conv_5x5_dil6 skip_connect"
).split()

# aggregation ops of a template: per-channel weighted sum / concat + 1x1 reduce
AGG_OP_NAMES = ["psum", "cat"]
