"""FPN (reference: models/fpn.py). NOTE the reference accumulates BOTTOM-UP: the finer lateral is bilinearly
resized to the next coarser level and added to that level's 1x1 lateral (fpn.py:51-56).  Here the add is the
`addend` of the lateral's implicit-GEMM epilogue and the ReLU of the 3x3 output conv is its epilogue."""
from torch import nn

import os

from . import ops
from .config import cfg

FPN_BLOCK = bool(int(os.environ.get("PRN_FPN_BLOCK", "1")))      # 0: the operator-by-operator level also at inference (cross-check)


class FPN(nn.Module):
    def __init__(self, in_channels, start_level=0):
        super().__init__()
        assert isinstance(in_channels, list)
        self.in_channels = in_channels
        self.out_channels = cfg.fpn.num_features
        self.num_ins = len(in_channels)
        self.backbone_end_level = self.num_ins
        self.start_level = start_level
        self.lateral_convs = nn.ModuleList(nn.Conv2d(in_channels[i], self.out_channels, 1)
                                           for i in range(start_level, self.backbone_end_level))
        self.fpn_convs = nn.ModuleList(nn.Conv2d(self.out_channels, self.out_channels, 3, padding=1)
                                       for _ in range(start_level, self.backbone_end_level))
        if cfg.fpn.high_level_mode == "retina":
            self.downsample_layers = nn.ModuleList(nn.Conv2d(self.out_channels, self.out_channels, 3, padding=1, stride=2)
                                                   for _ in range(2))
        self.interpolation_mode = cfg.fpn.interpolation_mode
        self.relu_pred_layers = cfg.fpn.relu_pred_layers
        self.high_level_mode = cfg.fpn.high_level_mode
        if self.interpolation_mode != "bilinear":
            raise NotImplementedError("fpn.interpolation_mode=%r" % self.interpolation_mode)

    def forward(self, inputs, return_inputs=False):
        """return_inputs: also hand the (forked) input features back, for the next consumer of the backbone features --
        their gradient then joins this module's lateral-conv input gradient in the GEMM epilogue (ops.conv2d_fork)."""
        assert len(inputs) == len(self.in_channels)
        import torch
        if not torch.is_grad_enabled() and inputs[0].is_cuda and self.high_level_mode not in ("original", "retina") and FPN_BLOCK:
            # inference: one C call per level (include/prn.h: prn_fpn_level_fwd)
            outs, prev = [], None
            for i, (lat, c) in enumerate(zip(self.lateral_convs, self.fpn_convs)):
                prev, p = ops.fpn_level(inputs[i + self.start_level], lat.weight, lat.bias, prev, c.weight, c.bias, self.relu_pred_layers)
                outs.append(p)
            return (outs, list(inputs)) if return_inputs else outs
        laterals, prev, idents = [], None, list(inputs)
        for i, lat in enumerate(self.lateral_convs):
            f = inputs[i + self.start_level]
            add = None
            if prev is not None:
                # the lateral also feeds its level's 3x3 conv: that gradient joins the resize's backward kernel (resize_bilinear_fork)
                add, laterals[-1] = ops.resize_bilinear_fork(prev, f.shape[2:])
            if return_inputs:
                prev, idents[i + self.start_level] = ops.conv2d_fork(f, lat.weight, lat.bias, addend=add)
            else:
                prev = ops.conv2d(f, lat.weight, lat.bias, addend=add)
            laterals.append(prev)
        epi = ops.EPI_RELU if self.relu_pred_layers else ops.EPI_NONE
        outs = [ops.conv2d(l, c.weight, c.bias, pad=1, epilogue=epi) for l, c in zip(laterals, self.fpn_convs)]
        if self.high_level_mode == "original":
            outs.append(outs[-1][:, :, ::2, ::2])            # max_pool2d(kernel 1, stride 2)
        elif self.high_level_mode == "retina":
            p6 = ops.conv2d(outs[-1], self.downsample_layers[0].weight, self.downsample_layers[0].bias, stride=2, pad=1)
            p7 = ops.conv2d(p6.relu(), self.downsample_layers[1].weight, self.downsample_layers[1].bias, stride=2, pad=1)
            outs += [p6, p7]
        return (outs, idents) if return_inputs else outs
