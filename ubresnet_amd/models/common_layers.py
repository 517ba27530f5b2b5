"""Building blocks with the reference's names, constructor signatures and state_dict keys
(models/common_layers.py).  They own parameters; the enclosing network's HIP graph executor
(ubresnet_amd/engine.py) runs them fused.
"""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:          # allow `import common_layers` with only this directory on sys.path
    sys.path.insert(0, _ROOT)

# One module object for both import styles of the reference (`sys.path += $UBRESNET_MODELDIR; import common_layers`
# and `import ubresnet_amd.models.common_layers`): a top-level import re-binds itself to the package module.
if __name__ != "ubresnet_amd.models.common_layers":
    import importlib as _il
    sys.modules[__name__] = _il.import_module("ubresnet_amd.models.common_layers")

import torch.nn as nn  # noqa: E402

from ubresnet_amd import nn_params as P  # noqa: E402

_STANDALONE = ("ubresnet_amd: %s is executed by the fused HIP graph of UResNet / ASPP_ResNet and cannot be called "
               "on its own; run the enclosing model (there is no eager PyTorch fallback)")


def conv3x3(in_planes, out_planes, stride=1):
    """3x3 convolution with padding (models/common_layers.py:13-15)"""
    return P.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


class BasicBlock(nn.Module):
    """models/common_layers.py:18-58"""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1):
        super(BasicBlock, self).__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = P.BatchNorm2d(planes)
        self.relu1 = P.ReLU(inplace=False)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = P.BatchNorm2d(planes)
        self.relu2 = P.ReLU(inplace=False)
        self.stride = stride
        self.bypass = None
        self.bnpass = None
        if inplanes != planes or stride > 1:
            self.bypass = P.Conv2d(inplanes, planes, kernel_size=1, stride=stride, padding=0, bias=False)
            self.bnpass = P.BatchNorm2d(planes)
        self.relu = P.ReLU(inplace=False)

    def forward(self, x):
        raise RuntimeError(_STANDALONE % "BasicBlock")

    def _grad_completion_order(self, prefix):
        """parameters in the order backward finishes their gradients (conv1.weight last)"""
        out = [(prefix + "bn2.weight", self.bn2.weight), (prefix + "bn2.bias", self.bn2.bias)]
        if self.bypass is not None:
            out += [(prefix + "bnpass.weight", self.bnpass.weight), (prefix + "bnpass.bias", self.bnpass.bias)]
        out += [(prefix + "conv2.weight", self.conv2.weight),
                (prefix + "bn1.weight", self.bn1.weight), (prefix + "bn1.bias", self.bn1.bias)]
        if self.bypass is not None:
            out += [(prefix + "bypass.weight", self.bypass.weight)]
        out += [(prefix + "conv1.weight", self.conv1.weight)]
        return out


class Bottleneck(nn.Module):
    """models/common_layers.py:61-106 -- defined but never instantiated by the reference
    (DoubleResNet uses BasicBlock, :112-115); kept only so the name resolves."""

    def __init__(self, inplanes, planes, stride=1):
        super(Bottleneck, self).__init__()
        raise NotImplementedError("Bottleneck is unused by the reference networks and not part of the HIP path")


class DoubleResNet(nn.Module):
    """models/common_layers.py:109-120"""

    def __init__(self, inplanes, planes, stride=1):
        super(DoubleResNet, self).__init__()
        self.res1 = BasicBlock(inplanes, planes, stride)
        self.res2 = BasicBlock(planes, planes, 1)

    def forward(self, x):
        raise RuntimeError(_STANDALONE % "DoubleResNet")

    def _grad_completion_order(self, prefix):
        return self.res2._grad_completion_order(prefix + "res2.") + self.res1._grad_completion_order(prefix + "res1.")


class ConvTransposeLayer(nn.Module):
    """models/common_layers.py:122-132"""

    def __init__(self, deconv_inplanes, deconv_outplanes, res_outplanes):
        super(ConvTransposeLayer, self).__init__()
        self.deconv = P.ConvTranspose2d(deconv_inplanes, deconv_outplanes, kernel_size=4, stride=2, padding=1, bias=False)
        self.res = DoubleResNet(res_outplanes + deconv_outplanes, res_outplanes, stride=1)

    def forward(self, x, skip_x):
        raise RuntimeError(_STANDALONE % "ConvTransposeLayer")

    def _grad_completion_order(self, prefix):
        return self.res._grad_completion_order(prefix + "res.") + [(prefix + "deconv.weight", self.deconv.weight)]
