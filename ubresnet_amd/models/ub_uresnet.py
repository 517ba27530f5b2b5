"""U-ResNet with the reference's constructor signature, attribute tree and state_dict keys
(models/ub_uresnet.py:29-147), executed by hand-written HIP kernels on MI355X.

    UResNet(num_classes=3, input_channels=3, inplanes=16, final_conv_kernels=16, showsizes=False)
    model.forward(x) / model(x):  x float32 [B, input_channels, H, W] (H, W multiples of 32) on a
    ROCm device  ->  log-softmax [B, num_classes, H, W] float32, autograd-connected.

Compute precision: fp32 by default (parity path).  bf16/fp16 storage with fp32 accumulation is
selected with ``model.compute_dtype = torch.bfloat16`` or by running under
``torch.autocast("cuda", dtype=torch.bfloat16)``; parameters stay fp32 masters either way.
"""
import math
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

# One module object for both import styles of the reference (`sys.path += $UBRESNET_MODELDIR; import ub_uresnet`
# and `import ubresnet_amd.models.ub_uresnet`): a top-level import re-binds itself to the package module.
if __name__ != "ubresnet_amd.models.ub_uresnet":
    import importlib as _il
    sys.modules[__name__] = _il.import_module("ubresnet_amd.models.ub_uresnet")

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from ubresnet_amd import nn_params as P  # noqa: E402
from ubresnet_amd.models.common_layers import *  # noqa: E402,F401,F403  (the reference does `from common_layers import *`)
from ubresnet_amd.models.common_layers import DoubleResNet, ConvTransposeLayer  # noqa: E402
from ubresnet_amd.autograd_fn import run_network  # noqa: E402


class UResNet(nn.Module):

    def __init__(self, num_classes=3, input_channels=3, inplanes=16, final_conv_kernels=16, showsizes=False):
        self.inplanes = inplanes
        super(UResNet, self).__init__()
        self._showsizes = showsizes
        self.compute_dtype = None      # None: fp32 unless torch.autocast says otherwise

        # stem (models/ub_uresnet.py:41-44)
        self.conv1 = P.Conv2d(input_channels, self.inplanes, kernel_size=7, stride=1, padding=3, bias=True)
        self.bn1 = P.BatchNorm2d(self.inplanes)
        self.relu1 = P.ReLU(inplace=False)
        self.pool1 = P.MaxPool2d(3, stride=2, padding=1)

        self.enc_layer1 = self._make_encoding_layer(self.inplanes * 1, self.inplanes * 2, stride=1)
        self.enc_layer2 = self._make_encoding_layer(self.inplanes * 2, self.inplanes * 4, stride=2)
        self.enc_layer3 = self._make_encoding_layer(self.inplanes * 4, self.inplanes * 8, stride=2)
        self.enc_layer4 = self._make_encoding_layer(self.inplanes * 8, self.inplanes * 16, stride=2)
        self.enc_layer5 = self._make_encoding_layer(self.inplanes * 16, self.inplanes * 32, stride=2)

        self.dec_layer5 = self._make_decoding_layer(self.inplanes * 32, self.inplanes * 16, self.inplanes * 16)
        self.dec_layer4 = self._make_decoding_layer(self.inplanes * 16, self.inplanes * 8, self.inplanes * 8)
        self.dec_layer3 = self._make_decoding_layer(self.inplanes * 8, self.inplanes * 4, self.inplanes * 4)
        self.dec_layer2 = self._make_decoding_layer(self.inplanes * 4, self.inplanes * 2, self.inplanes * 2)
        self.dec_layer1 = self._make_decoding_layer(self.inplanes * 2, self.inplanes, self.inplanes)

        # head (models/ub_uresnet.py:59-70)
        self.nkernels = final_conv_kernels
        self.conv10 = P.Conv2d(self.inplanes, self.nkernels, kernel_size=7, stride=1, padding=3, bias=True)
        self.bn10 = P.BatchNorm2d(self.nkernels)
        self.relu10 = P.ReLU(inplace=False)
        self.conv11 = P.Conv2d(self.nkernels, num_classes, kernel_size=7, stride=1, padding=3, bias=True)
        self.softmax = P.LogSoftmax(dim=1)

        # initialisation (models/ub_uresnet.py:73-79)
        for m in self.modules():
            if isinstance(m, nn.Conv2d) or isinstance(m, nn.ConvTranspose2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

        if self.inplanes % 16 or self.nkernels % 16:
            raise ValueError("ubresnet_amd: inplanes and final_conv_kernels must be multiples of 16 "
                             "(MFMA channel fragment); got %d, %d" % (self.inplanes, self.nkernels))
        if not 1 <= input_channels <= 4:
            raise ValueError("ubresnet_amd: input_channels must be 1..4 (got %d)" % input_channels)
        if not 1 <= num_classes <= 16:
            raise ValueError("ubresnet_amd: num_classes must be 1..16 (got %d)" % num_classes)

    def _make_encoding_layer(self, inplanes, planes, stride=2):
        return DoubleResNet(inplanes, planes, stride=stride)

    def _make_decoding_layer(self, inplanes, deconvplanes, resnetplanes):
        return ConvTransposeLayer(inplanes, deconvplanes, resnetplanes)

    # order in which backward completes parameter gradients (flat gradient buffer layout)
    def _grad_completion_order(self):
        out = [("conv11.weight", self.conv11.weight), ("conv11.bias", self.conv11.bias),
               ("conv10.weight", self.conv10.weight), ("conv10.bias", self.conv10.bias),
               ("bn10.weight", self.bn10.weight), ("bn10.bias", self.bn10.bias)]
        for i in (1, 2, 3, 4, 5):
            out += getattr(self, "dec_layer%d" % i)._grad_completion_order("dec_layer%d." % i)
        for i in (5, 4, 3, 2, 1):
            out += getattr(self, "enc_layer%d" % i)._grad_completion_order("enc_layer%d." % i)
        out += [("bn1.weight", self.bn1.weight), ("bn1.bias", self.bn1.bias),
                ("conv1.weight", self.conv1.weight), ("conv1.bias", self.conv1.bias)]
        return out

    def forward(self, x):
        if self._showsizes:
            ip = self.inplanes
            B, _, H, W = x.shape
            print("input: ", x.size(), " is_cuda=", x.is_cuda)
            print("after conv1, x0: ", torch.Size((B, ip, H, W)))
            print("after encoding: ")
            for i in range(1, 6):
                print("  x%d: " % i, torch.Size((B, ip * 2 ** i, H >> i, W >> i)))
            print("after decoding:")
            for i in (5, 4, 3, 2, 1):
                print("  dec%d: " % i, torch.Size((B, ip * 2 ** (i - 1), H >> (i - 1), W >> (i - 1))), " iscuda=", x.is_cuda)
            print("  softmax: ", torch.Size((B, self.conv11.out_channels, H, W)))
        return run_network(self, "uresnet", x)
