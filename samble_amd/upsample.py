"""Drop-in `UpSampleInterpolation` (reference models/upsample.py:136-213), the decoder layer of the
shipped segmentation preset (`us_which: interpolation`, `distance_type: xyz`, K = 3): cross-set
K-nearest neighbours with reference-normalised distances run on the HIP kNN kernels
(`ops.select_neighbors_interpolate`), inverse-distance weights and the two Conv1d+BN+LeakyReLU
blocks are stock torch."""
from __future__ import annotations

import torch
from torch import nn

from . import ops


class UpSampleInterpolation(nn.Module):
    def __init__(self, config_upsample, layer):
        super().__init__()
        q_in = config_upsample.q_in[layer]
        v_out = config_upsample.v_out[layer]
        self.distance_type = config_upsample.interpolation.distance_type[layer]
        self.K = config_upsample.interpolation.K[layer]
        self.conv = nn.Sequential(nn.Conv1d(q_in, v_out, 1, bias=False), nn.BatchNorm1d(v_out),
                                  nn.LeakyReLU(negative_slope=0.2))
        self.res_conv = nn.Sequential(nn.Conv1d(2 * v_out, v_out, 1, bias=False), nn.BatchNorm1d(v_out),
                                      nn.LeakyReLU(negative_slope=0.2))

    def forward(self, pcd_up, pcd_down, pcd_up_xyz):
        (points_select, idx_select, points_select_xyz), (points_drop, idx_drop) = pcd_down
        interpolated_points = self.interpolate(pcd_up, points_select, pcd_up_xyz, points_select_xyz,
                                               distance_type=self.distance_type, K=self.K)
        x = torch.concat([pcd_up, interpolated_points], dim=1)
        return self.res_conv(x)

    def interpolate(self, pcd_up, points_select, pcd_up_xyz, points_select_xyz, distance_type="feature", K=3):
        points_select_conv = self.conv(points_select)
        if distance_type == "xyz":
            neighbors, _, d_neighbors = ops.select_neighbors_interpolate(pcd_up_xyz, points_select_xyz,
                                                                         points_select_conv, K=K)
        elif distance_type == "feature":
            # the reference back-propagates through cdist of the (normalised) features: the neighbour SEARCH
            # runs on the HIP kNN kernels (indices carry no gradient), the K distances per point are then
            # recomputed differentiably for the selected pairs only (utils/ops.py:17-44 normalisation)
            _, idx, _ = ops.select_neighbors_interpolate(pcd_up.detach(), points_select.detach(),
                                                         points_select_conv.detach(), K=K)
            a = pcd_up.permute(0, 2, 1)
            b = points_select.permute(0, 2, 1)
            a_mean = torch.mean(a, dim=1, keepdim=True)
            a, b = a - a_mean, b - a_mean
            a_std = torch.mean(torch.std(a, dim=1, keepdim=True), dim=2, keepdim=True)
            a, b = a / a_std, b / a_std
            d_neighbors = torch.linalg.vector_norm(a.unsqueeze(2) - ops.index_points(b, idx), dim=-1)  # (B,N,K)
            neighbors = ops.index_points(points_select_conv.permute(0, 2, 1), idx).permute(0, 3, 1, 2)
        else:
            raise ValueError(f"upsample interpolation distance type can only be feature or xyz! Got: {distance_type}")
        weights = 1.0 / (d_neighbors + 1e-8)
        weights = weights / torch.sum(weights, dim=-1, keepdim=True)
        return torch.sum(neighbors * weights.unsqueeze(dim=1), dim=-1)


def upsample_config(preset: str = "seg"):
    from .config import to_attr
    return to_attr(dict(us_which="interpolation", interpolation=dict(distance_type=["xyz", "xyz"], K=[3, 3]),
                        q_in=[128, 128], q_out=[128, 128], k_in=[128, 128], k_out=[128, 128], v_in=[128, 128],
                        v_out=[128, 128], num_heads=[4, 4]))
