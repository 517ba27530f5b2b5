"""ASPP-ResNet with the reference's constructor signature, attribute tree and state_dict keys
(models/ASPP_ResNet.py:188-523), executed by hand-written HIP kernels on MI355X.

    ASPP_ResNet(num_classes=3, in_channels=3, inplanes=16, showsizes=True)

U-ResNet encoder; for encoder levels 3,4,5 an atrous-spatial-pyramid block (1x1, 3x3 d1, 3x3 d3,
3x3 d5 convs -> BN -> ReLU, 16 channels each, plus MaxPool2d(3,1,1) of the input; :227-263)
followed by a 1x1 conv+BN+ReLU back to C channels (:280-286) is concatenated in front of the
encoder output (:459,471,483) and feeds a wider decoder (:361-375).

As in the reference the network is only consistent for inplanes=16 (the ASPP branch width is
hard-wired to 16 so `inplanes*12/20/36` at :343-351 only match then); other values raise.
The reference file also imports `commands`, ROOT, larcv and torchvision (:22,31,32,45-47); none of
them is used by the model, so they are not imported here.  `ASPP_ResNet1` (the module name the
reference's train scripts import, Sem_Seg_ASPP_ResNet1.py:43) is the alias module ASPP_ResNet1.py next to this file.
"""
import math
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

if __name__ != "ubresnet_amd.models.ASPP_ResNet":
    import importlib as _il
    sys.modules[__name__] = _il.import_module("ubresnet_amd.models.ASPP_ResNet")

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from ubresnet_amd import nn_params as P  # noqa: E402
from ubresnet_amd.models.common_layers import (BasicBlock, Bottleneck, ConvTransposeLayer, DoubleResNet,  # noqa: E402,F401
                                               conv3x3, _STANDALONE)
from ubresnet_amd.autograd_fn import run_network  # noqa: E402


class ASPP(nn.Module):
    """models/ASPP_ResNet.py:188-263"""

    def __init__(self, inplanes, outplanes=16, nkernels=16, showsizes=False):
        super(ASPP, self).__init__()
        stride = 1
        self.inplanes = inplanes
        self.outplanes = outplanes
        self.nkernels = nkernels
        self.showsizes = showsizes
        self.B1_conv = P.Conv2d(self.inplanes, self.outplanes, kernel_size=1, stride=stride, padding=0, dilation=1, bias=True)
        self.B1_bn = P.BatchNorm2d(self.nkernels)
        self.B1_relu = P.ReLU(inplace=True)
        self.B2_conv = P.Conv2d(self.inplanes, self.outplanes, kernel_size=3, stride=stride, padding=1, dilation=1, bias=True)
        self.B2_bn = P.BatchNorm2d(self.nkernels)
        self.B2_relu = P.ReLU(inplace=True)
        self.B3_conv = P.Conv2d(self.inplanes, self.outplanes, kernel_size=3, stride=stride, padding=3, dilation=3, bias=True)
        self.B3_bn = P.BatchNorm2d(self.nkernels)
        self.B3_relu = P.ReLU(inplace=True)
        self.B4_conv = P.Conv2d(self.inplanes, self.outplanes, kernel_size=3, stride=stride, padding=5, dilation=5, bias=True)
        self.B4_bn = P.BatchNorm2d(self.nkernels)
        self.B4_relu = P.ReLU(inplace=True)
        self.B5_gp = P.MaxPool2d(kernel_size=3, stride=stride, padding=1, dilation=1, return_indices=False, ceil_mode=False)
        if outplanes != 16 or nkernels != 16:
            raise ValueError("ubresnet_amd: ASPP branch width is 16 in the reference network")

    def forward(self, x):
        raise RuntimeError(_STANDALONE % "ASPP")

    def branches(self):
        """(conv, bn, kernel, dilation) of the four conv branches in concat order"""
        return [(self.B1_conv, self.B1_bn, 1, 1), (self.B2_conv, self.B2_bn, 3, 1),
                (self.B3_conv, self.B3_bn, 3, 3), (self.B4_conv, self.B4_bn, 3, 5)]

    def _grad_completion_order(self, prefix):
        out = []
        for i, (conv, bn, _, _) in enumerate(self.branches()):
            b = "B%d" % (i + 1)
            out += [(prefix + b + "_bn.weight", bn.weight), (prefix + b + "_bn.bias", bn.bias),
                    (prefix + b + "_conv.weight", conv.weight), (prefix + b + "_conv.bias", conv.bias)]
        return out


class ASPP_post(nn.Module):
    """models/ASPP_ResNet.py:266-286"""

    def __init__(self, inplanes, outplanes):
        super(ASPP_post, self).__init__()
        self.inplanes = inplanes
        self.outplanes = outplanes
        self.nkernels = outplanes
        self.ASPP_conv = P.Conv2d(self.inplanes, self.outplanes, kernel_size=1, stride=1, padding=0, bias=True)
        self.ASPP_bn = P.BatchNorm2d(self.nkernels)
        self.ASPP_relu = P.ReLU(inplace=True)

    def forward(self, x):
        raise RuntimeError(_STANDALONE % "ASPP_post")

    def _grad_completion_order(self, prefix):
        return [(prefix + "ASPP_bn.weight", self.ASPP_bn.weight), (prefix + "ASPP_bn.bias", self.ASPP_bn.bias),
                (prefix + "ASPP_conv.weight", self.ASPP_conv.weight), (prefix + "ASPP_conv.bias", self.ASPP_conv.bias)]


class ASPP_ResNet(nn.Module):

    def __init__(self, num_classes=3, in_channels=3, inplanes=16, showsizes=True):
        self.inplanes = inplanes
        super(ASPP_ResNet, self).__init__()
        if inplanes != 16:
            raise ValueError("ASPP_ResNet is only consistent for inplanes=16 (models/ASPP_ResNet.py:343-351); got %d" % inplanes)
        self.nkernels = 16
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.showsizes = showsizes
        self.compute_dtype = None

        self.conv1 = P.Conv2d(in_channels, self.inplanes, kernel_size=7, stride=1, padding=3, bias=True)
        self.bn1 = P.BatchNorm2d(self.inplanes)
        self.relu1 = P.ReLU(inplace=True)
        self.pool1 = P.MaxPool2d(3, stride=2, padding=1)

        self.enc_layer1 = self._make_encoding_layer(self.inplanes * 1, self.inplanes * 2, stride=1)
        self.enc_layer2 = self._make_encoding_layer(self.inplanes * 2, self.inplanes * 4, stride=2)
        self.enc_layer3 = self._make_encoding_layer(self.inplanes * 4, self.inplanes * 8, stride=2)
        self.enc_layer4 = self._make_encoding_layer(self.inplanes * 8, self.inplanes * 16, stride=2)
        self.enc_layer5 = self._make_encoding_layer(self.inplanes * 16, self.inplanes * 32, stride=2)

        self.ASPP_layer_enc3 = self.ASPP_layer(self.inplanes * 8)
        self.ASPP_combine_enc3 = self.ASPP_combine(self.inplanes * 12, self.inplanes * 8)
        self.ASPP_layer_enc4 = self.ASPP_layer(self.inplanes * 16)
        self.ASPP_combine_enc4 = self.ASPP_combine(self.inplanes * 20, self.inplanes * 16)
        self.ASPP_layer_enc5 = self.ASPP_layer(self.inplanes * 32)
        self.ASPP_combine_enc5 = self.ASPP_combine(self.inplanes * 36, self.inplanes * 32)

        self.dec_layer5 = self._make_decoding_layer(self.inplanes * 64, self.inplanes * 16, self.inplanes * 32)
        self.dec_layer4 = self._make_decoding_layer(self.inplanes * 32, self.inplanes * 8, self.inplanes * 16)
        self.dec_layer3 = self._make_decoding_layer(self.inplanes * 16, self.inplanes * 4, self.inplanes * 4)
        self.dec_layer2 = self._make_decoding_layer(self.inplanes * 4, self.inplanes * 2, self.inplanes * 2)
        self.dec_layer1 = self._make_decoding_layer(self.inplanes * 2, self.inplanes, self.inplanes)

        self.nkernels = 16
        self.conv10 = P.Conv2d(self.inplanes, self.nkernels, kernel_size=7, stride=1, padding=3, bias=True)
        self.bn10 = P.BatchNorm2d(self.nkernels)
        self.relu10 = P.ReLU(inplace=True)
        self.conv11 = P.Conv2d(self.inplanes, num_classes, kernel_size=7, stride=1, padding=3, bias=True)
        self.softmax = P.LogSoftmax(dim=1)

        for m in self.modules():
            if isinstance(m, nn.Conv2d) or isinstance(m, nn.ConvTranspose2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        if not 1 <= in_channels <= 4:
            raise ValueError("ubresnet_amd: in_channels must be 1..4 (got %d)" % in_channels)
        if not 1 <= num_classes <= 16:
            raise ValueError("ubresnet_amd: num_classes must be 1..16 (got %d)" % num_classes)

    def _make_encoding_layer(self, inplanes, planes, stride=2):
        return DoubleResNet(inplanes, planes, stride=stride)

    def _make_decoding_layer(self, inplanes, deconvplanes, resnetplanes):
        return ConvTransposeLayer(inplanes, deconvplanes, resnetplanes)

    def ASPP_layer(self, inplanes):
        return ASPP(inplanes)

    def ASPP_combine(self, inplanes, outplanes):
        return ASPP_post(inplanes, outplanes)

    def _grad_completion_order(self):
        out = [("conv11.weight", self.conv11.weight), ("conv11.bias", self.conv11.bias),
               ("conv10.weight", self.conv10.weight), ("conv10.bias", self.conv10.bias),
               ("bn10.weight", self.bn10.weight), ("bn10.bias", self.bn10.bias)]
        for i in (1, 2, 3, 4, 5):
            out += getattr(self, "dec_layer%d" % i)._grad_completion_order("dec_layer%d." % i)
        for i in (5, 4, 3):
            out += getattr(self, "ASPP_combine_enc%d" % i)._grad_completion_order("ASPP_combine_enc%d." % i)
            out += getattr(self, "ASPP_layer_enc%d" % i)._grad_completion_order("ASPP_layer_enc%d." % i)
            out += getattr(self, "enc_layer%d" % i)._grad_completion_order("enc_layer%d." % i)
        for i in (2, 1):
            out += getattr(self, "enc_layer%d" % i)._grad_completion_order("enc_layer%d." % i)
        out += [("bn1.weight", self.bn1.weight), ("bn1.bias", self.bn1.bias),
                ("conv1.weight", self.conv1.weight), ("conv1.bias", self.conv1.bias)]
        return out

    def forward(self, x):
        if self.showsizes:
            ip = self.inplanes
            B, _, H, W = x.shape
            print("x_in dim:", x.size())
            print("x0 dim:", torch.Size((B, ip, H, W)))
            for i in range(1, 6):
                print("e%d dim:" % i, torch.Size((B, ip * 2 ** i, H >> i, W >> i)))
            for i in (3, 4, 5):
                c = ip * 2 ** i
                print("e%d_ASPP dim:" % i, torch.Size((B, 64 + c, H >> i, W >> i)), "-> e%d_skip dim:" % i, torch.Size((B, 2 * c, H >> i, W >> i)))
            for i, c in ((5, 32 * ip), (4, 16 * ip), (3, 4 * ip), (2, 2 * ip), (1, ip)):
                print("d%d dim:" % i, torch.Size((B, c, H >> (i - 1), W >> (i - 1))))
            print("softmax dim:", torch.Size((B, self.num_classes, H, W)))
        return run_network(self, "aspp", x)
