"""ResNet-DCN backbone (reference: models/backbone.py).

Module tree, parameter names and the plugin contract (`.channels`, `.layers`, `.backbone_modules`,
`.init_backbone`, `.add_layer`, `forward -> tuple`, `construct_backbone(cfg.backbone)`) are the reference's;
nn.Conv2d / nn.BatchNorm2d instances are used as parameter containers only -- the arithmetic runs in the HIP
operators (planerecnet_amd.ops): every BatchNorm launch also applies the ReLU and, for bn3, the residual add.
"""
import os

import torch
from torch import nn

from . import blocks, ops
from .dcn import DeformableConv2d


STAGE_FORK = os.environ.get("PRN_STAGE_FORK", "1") == "1"      # 0: the stage outputs' gradients meet in autograd's accumulation pass (A/B)


def _bn(m, x, residual=None, relu=False):
    return ops.batch_norm_module(m, x, residual, relu)


def folded_bn(conv_w, conv_b, m):
    """Inference only (eval-mode BatchNorm, no autograd): scale the conv weights by gamma/sqrt(var+eps) and fold the shift
    into a bias, so that conv + BN (+ residual + ReLU) is ONE GEMM launch with a fused epilogue and the BatchNorm kernels
    disappear from the forward pass.  Cached until any of the five tensors changes (their autograd version counters)."""
    key = (conv_w._version, None if conv_b is None else conv_b._version, m.weight._version, m.bias._version, m.running_mean._version,
           m.running_var._version, conv_w.data_ptr())
    c = m.__dict__.get("_prn_folded")
    if c is None or c[0] != key:
        with torch.no_grad():
            scale = m.weight * torch.rsqrt(m.running_var + m.eps)
            w = (conv_w * scale.view(-1, 1, 1, 1)).contiguous()
            b = m.bias - m.running_mean * scale if conv_b is None else (conv_b - m.running_mean) * scale + m.bias
        if c is not None:
            ops._split_drop(c[1].data_ptr())                   # (kept split-GEMM images of the weights this replaces)
        c = m.__dict__["_prn_folded"] = (key, w, b.contiguous())
        ops._stamp(w)                                          # a derived operand that persists: its cut images may be kept (ops.split_images)
    return c[1], c[2]


def can_fold(m):
    return (not m.training) and (not torch.is_grad_enabled())


def conv_bn(x, conv, m, stride=1, pad=0, relu=False, residual=None, in_mode=ops.IN_ZERO):
    """conv -> BatchNorm (-> + residual) (-> ReLU): folded into one launch at inference, two ops otherwise."""
    if can_fold(m):
        w, b = folded_bn(conv.weight, conv.bias, m)
        return ops.conv2d(x, w, b, stride, pad, in_mode, ops.EPI_RELU if relu else ops.EPI_NONE, residual)
    return _bn(m, ops.conv2d(x, conv.weight, conv.bias, stride, pad, in_mode), residual, relu)


class Bottleneck(nn.Module):
    """1x1 -> 3x3 (stride here; plain or deformable) -> 1x1, each + BN, residual, ReLU (backbone.py:5-73)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, norm_layer=nn.BatchNorm2d, dilation=1, use_dcn=False):
        super().__init__()
        if dilation != 1:
            raise NotImplementedError("atrous layers are not used by the PlaneRecNet configs")
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = (DeformableConv2d(planes, planes, 3, stride=stride, padding=1, bias=True) if use_dcn
                      else nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False))
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = norm_layer(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def _apply(self, fn, recurse=True):                    # .to() / .cuda() / .float() replace parameter and buffer objects: drop what blocks.py derived from them
        r = super()._apply(fn, recurse)
        blocks.invalidate(self)
        return r

    def forward(self, x, hand_back=False):
        """hand_back (training path, blocks with a downsample branch): also return the block's INPUT as handed on by the downsample convolution's fork --
        whoever else reads that tensor (the FPN / the depth decoder read the stage outputs) takes it from there, so that its gradient joins the
        downsample convolution's input gradient inside that launch instead of in autograd's separate accumulation pass."""
        if can_fold(self.bn1):                                 # inference: every conv + BN (+ residual + ReLU) is one launch
            out = conv_bn(x, self.conv1, self.bn1, relu=True)
            if isinstance(self.conv2, DeformableConv2d):
                out = self.conv2(out, fold=self.bn2)
            else:
                out = conv_bn(out, self.conv2, self.bn2, self.stride, 1, relu=True)
            res = x if self.downsample is None else conv_bn(x, self.downsample[0], self.downsample[1], self.downsample[0].stride[0])
            out = conv_bn(out, self.conv3, self.bn3, relu=True, residual=res)
            return (out, x) if hand_back else out
        if blocks.usable(self, x):                             # training mode on the device: the whole block is ONE call each way (blocks.py, include/prn.h: prn_bottleneck_*)
            return blocks.bottleneck_train(self, x, hand_back)
        # operator by operator (frozen BatchNorm, profiler-bracketed runs, shapes the block entry points do not take): every operator writes its own result.
        # x has a second consumer (the identity branch or the downsample conv): hand it on through the fork so that both
        # gradients of x meet in conv1's input-gradient epilogue instead of in a separate accumulation kernel
        out, x = ops.conv2d_fork(x, self.conv1.weight)
        out = _bn(self.bn1, out, relu=True)
        if isinstance(self.conv2, DeformableConv2d):
            out = self.conv2(out)
        else:
            out = ops.conv2d(out, self.conv2.weight, stride=self.stride, pad=1)
        out = _bn(self.bn2, out, relu=True)
        out = ops.conv2d(out, self.conv3.weight)
        res = x
        if self.downsample is not None:
            if hand_back:
                r, x = ops.conv2d_fork(x, self.downsample[0].weight, stride=self.downsample[0].stride[0])
            else:
                r = ops.conv2d(x, self.downsample[0].weight, stride=self.downsample[0].stride[0])
            res = _bn(self.downsample[1], r)
        out = _bn(self.bn3, out, residual=res, relu=True)
        return (out, x) if hand_back else out


class ResNetBackbone(nn.Module):
    def __init__(self, layers, dcn_layers=[0, 0, 0, 0], dcn_interval=1, atrous_layers=[], block=Bottleneck,
                 norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.num_base_layers = len(layers)
        self.layers = nn.ModuleList()
        self.channels = []
        self.norm_layer = norm_layer
        self.dilation = 1
        self.atrous_layers = atrous_layers
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        for i, (planes, n) in enumerate(zip((64, 128, 256, 512), layers)):
            self._make_layer(block, planes, n, stride=1 if i == 0 else 2, dcn_layers=dcn_layers[i], dcn_interval=dcn_interval)
        # modules that a pretrained ImageNet checkpoint initialises (read by PlaneRecNet.init_weights)
        self.backbone_modules = [m for m in self.modules() if isinstance(m, nn.Conv2d)]

    def _make_layer(self, block, planes, blocks, stride=1, dcn_layers=0, dcn_interval=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            if len(self.layers) in self.atrous_layers:
                self.dilation += 1
                stride = 1
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride=stride, bias=False),
                                       self.norm_layer(planes * block.expansion))
        # DCN placement rule: backbone.py:170,184
        stage = [block(self.inplanes, planes, stride, downsample, self.norm_layer, self.dilation, use_dcn=dcn_layers >= blocks)]
        self.inplanes = planes * block.expansion
        for i in range(1, blocks):
            stage.append(block(self.inplanes, planes, norm_layer=self.norm_layer,
                               use_dcn=((i + dcn_layers) >= blocks) and (i % dcn_interval == 0)))
        layer = nn.Sequential(*stage)
        self.channels.append(planes * block.expansion)
        self.layers.append(layer)
        return layer

    def forward(self, x):
        x = conv_bn(x, self.conv1, self.bn1, 2, 3, relu=True)
        x = ops.max_pool_3x3_s2(x)
        outs = []
        for layer in self.layers:
            blocks = list(layer)
            if STAGE_FORK and outs and isinstance(blocks[0], Bottleneck) and blocks[0].downsample is not None:
                # the previous stage's output has readers outside the backbone (FPN, depth decoder): they get it from the first block's fork
                x, outs[-1] = blocks[0](x, hand_back=True)
                blocks = blocks[1:]
            for block in blocks:
                x = block(x)
            outs.append(x)
        return tuple(outs)

    def init_backbone(self, path):
        """Load torchvision-style ImageNet weights: layerN.* -> layers.(N-1).*, strict=False (backbone.py:211-224)."""
        sd = torch.load(path, map_location="cpu")
        for key in list(sd):
            if key.startswith("layer"):
                sd["layers." + str(int(key[5]) - 1) + key[6:]] = sd.pop(key)
        self.load_state_dict(sd, strict=False)

    def add_layer(self, conv_channels=1024, downsample=2, depth=1, block=Bottleneck):
        self._make_layer(block, conv_channels // block.expansion, blocks=depth, stride=downsample)


def construct_backbone(cfg):
    """cfg.type(*cfg.args) -- the backbone plugin hook (backbone.py:233-243)."""
    backbone = cfg.type(*cfg.args)
    while len(backbone.layers) < max(cfg.selected_layers) + 1:
        backbone.add_layer()
    return backbone
