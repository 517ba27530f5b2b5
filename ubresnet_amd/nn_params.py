"""Parameter-holder layers.

The mirror modules keep torch.nn layer objects so that ``state_dict`` keys, default
initialisation, ``print(model)``, ``.to()``/``.cuda()`` and checkpoint loading are exactly the
reference's.  They are CONTAINERS only: the HIP graph executor
(ubresnet_amd/engine.py) reads their parameters; calling them would run PyTorch's own kernels,
so ``forward`` raises instead of silently falling back.
"""
import torch.nn as nn

_MSG = ("ubresnet_amd: %s is a parameter holder; the arithmetic runs in the HIP graph executor of the enclosing "
        "UResNet / ASPP_ResNet module (no eager PyTorch fallback)")


class Conv2d(nn.Conv2d):
    def forward(self, *a, **k):
        raise RuntimeError(_MSG % "Conv2d")


class ConvTranspose2d(nn.ConvTranspose2d):
    def forward(self, *a, **k):
        raise RuntimeError(_MSG % "ConvTranspose2d")


class BatchNorm2d(nn.BatchNorm2d):
    def forward(self, *a, **k):
        raise RuntimeError(_MSG % "BatchNorm2d")


class ReLU(nn.ReLU):
    def forward(self, *a, **k):
        raise RuntimeError(_MSG % "ReLU")


class MaxPool2d(nn.MaxPool2d):
    def forward(self, *a, **k):
        raise RuntimeError(_MSG % "MaxPool2d")


class LogSoftmax(nn.LogSoftmax):
    def forward(self, *a, **k):
        raise RuntimeError(_MSG % "LogSoftmax")
