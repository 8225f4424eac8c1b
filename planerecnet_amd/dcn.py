"""Modulated deformable convolution module (reference: models/dcn.py:11-67).

Same parameters / state-dict keys as the reference module (`offset_conv`, `modulator_conv`, `regular_conv`;
zero-initialised offset and modulator convs, dcn.py:32-43).  Forward differs in *how* it is computed, not
in what: the 18-channel offset conv and the 9-channel modulator conv read the same input with the same
geometry, so they run as ONE 27-channel implicit-GEMM launch; the clamp(+-max(h,w)/4) and 2*sigmoid are
folded into the gather table of the fused operator (prn_dcnv2_table), and the bilinear sampling is the operand loader of
the MFMA contraction (prn_dcnv2_fwd) -- no column tensor.  `ops.deform_conv2d` is the drop-in for the reference's
`torchvision.ops.deform_conv2d` call (same signature); `forward_reference_form` below calls it exactly as dcn.py:52-67 does.
"""
import torch
from torch import nn

from . import ops


class DeformableConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=False):
        super().__init__()
        if kernel_size != 3 or padding != 1:
            raise NotImplementedError("the hot path only uses 3x3 / pad 1 deformable convolutions")
        self.stride, self.padding = stride, padding
        self.offset_conv = nn.Conv2d(in_channels, 2 * 9, 3, stride=stride, padding=1, bias=True)
        self.modulator_conv = nn.Conv2d(in_channels, 9, 3, stride=stride, padding=1, bias=True)
        for m in (self.offset_conv, self.modulator_conv):
            nn.init.zeros_(m.weight)
            nn.init.zeros_(m.bias)
        self.regular_conv = nn.Conv2d(in_channels, out_channels, 3, stride=stride, padding=1, bias=bias)
        self._link_offset_storage()

    # The 18-channel offset conv and the 9-channel modulator conv run as ONE 27-channel GEMM.  Their four parameters stay the
    # reference's (state-dict keys, optimizer entries), but their storage is one [27, C, 3, 3] / [27] block of which they are
    # views: no per-step concatenation of the weights and no split of the gradients (22 + 22 small launches per step before).
    def _link_offset_storage(self):
        with torch.no_grad():
            w = torch.cat([self.offset_conv.weight.data, self.modulator_conv.weight.data], 0).contiguous()
            b = torch.cat([self.offset_conv.bias.data, self.modulator_conv.bias.data], 0).contiguous()
            self.offset_conv.weight.data, self.modulator_conv.weight.data = w[:18], w[18:]
            self.offset_conv.bias.data, self.modulator_conv.bias.data = b[:18], b[18:]
        self.__dict__["_w27"], self.__dict__["_b27"] = w, b

    def _merged(self):
        w, b = self.__dict__["_w27"], self.__dict__["_b27"]
        ow, mw, ob, mb = self.offset_conv.weight, self.modulator_conv.weight, self.offset_conv.bias, self.modulator_conv.bias
        if (ow.data_ptr() != w.data_ptr() or mw.data_ptr() != w.data_ptr() + 18 * w[0].numel() * w.element_size()
                or ob.data_ptr() != b.data_ptr() or mb.data_ptr() != b.data_ptr() + 18 * b.element_size() or ow.device != w.device):
            self._link_offset_storage()                    # someone replaced a parameter's storage (module.to(), manual .data assignment)
            w, b = self.__dict__["_w27"], self.__dict__["_b27"]
        return w, b

    def _apply(self, fn, recurse=True):                    # .to() / .cuda() / .float(): re-establish the shared storage afterwards
        r = super()._apply(fn, recurse)
        self._link_offset_storage()
        return r

    def forward(self, x, fold=None):
        """fold: an eval-mode BatchNorm2d to fold into the contraction (inference; the output is then ReLU'd as well)."""
        h, w = x.shape[2:]
        w27, b27 = self._merged()
        if fold is not None:
            from .backbone import folded_bn
            wf, bf = folded_bn(self.regular_conv.weight, self.regular_conv.bias, fold)
            om = ops.conv2d(x, w27, b27, stride=self.stride, pad=1)
            return ops.deform_conv2d_raw_relu(x, om, wf, bf, self.stride, max(h, w) / 4.0)
        return ops.deform_conv_block(x, self.offset_conv.weight, self.modulator_conv.weight, self.offset_conv.bias, self.modulator_conv.bias, w27, b27,
                                     self.regular_conv.weight, self.regular_conv.bias, self.stride, max(h, w) / 4.0)

    def forward_reference_form(self, x):
        """The reference's forward, statement by statement (models/dcn.py:52-67), on the drop-in operator: three separate
        ops with the torchvision call signature.  Same result as forward(); kept as the binding example and parity check."""
        h, w = x.shape[2:]
        max_offset = max(h, w) / 4.0
        offset = ops.conv2d(x, self.offset_conv.weight, self.offset_conv.bias, stride=self.stride, pad=1).clamp(-max_offset, max_offset)
        modulator = 2.0 * torch.sigmoid(ops.conv2d(x, self.modulator_conv.weight, self.modulator_conv.bias, stride=self.stride, pad=1))
        return ops.deform_conv2d(input=x, offset=offset, weight=self.regular_conv.weight, bias=self.regular_conv.bias,
                                 padding=self.padding, mask=modulator, stride=self.stride)
