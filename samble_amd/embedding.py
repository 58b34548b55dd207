"""Drop-in `EdgeConv` (reference models/embedding.py:7-39): the neighbour build (`ops.group`) runs on
the HIP kNN kernels (exact (a-b)^2 path for xyz, fused MFMA Gram + top-K for 64-d features); the two
1x1 Conv2d + BatchNorm2d + LeakyReLU blocks and the max over K are stock torch."""
from __future__ import annotations

from torch import nn

from . import ops


class EdgeConv(nn.Module):
    def __init__(self, config_embedding, layer):
        super().__init__()
        self.K = config_embedding.K[layer]
        self.group_type = config_embedding.group_type[layer]
        self.normal_channel = config_embedding.normal_channel
        c1_in, c1_out = config_embedding.conv1_in[layer], config_embedding.conv1_out[layer]
        c2_in, c2_out = config_embedding.conv2_in[layer], config_embedding.conv2_out[layer]
        self.conv1 = nn.Sequential(nn.Conv2d(c1_in, c1_out, kernel_size=1, bias=False), nn.BatchNorm2d(c1_out),
                                   nn.LeakyReLU(negative_slope=0.2))
        self.conv2 = nn.Sequential(nn.Conv2d(c2_in, c2_out, kernel_size=1, bias=False), nn.BatchNorm2d(c2_out),
                                   nn.LeakyReLU(negative_slope=0.2))

    def forward(self, x):
        x, _ = ops.group(x, self.K, self.group_type, self.normal_channel)
        x = self.conv1(x)
        x = self.conv2(x)
        return x.max(dim=-1, keepdim=False)[0]


def embedding_config(preset: str = "cls"):
    """`config.feature_learning_block.embedding` of the shipped presets (two EdgeConv layers)."""
    from .config import to_attr
    return to_attr(dict(K=[32, 32], group_type=["center_diff", "center_diff"], normal_channel=False,
                        conv1_in=[6, 128], conv1_out=[64, 64], conv2_in=[64, 64], conv2_out=[64, 64]))
