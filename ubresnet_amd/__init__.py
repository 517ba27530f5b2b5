"""MI355X-native U-ResNet semantic-segmentation path (drop-in for NuTufts/ubresnet models).

Public surface mirrors the reference's modules:
  ubresnet_amd.models.ub_uresnet.UResNet            (models/ub_uresnet.py:29)
  ubresnet_amd.models.ASPP_ResNet.ASPP_ResNet       (models/ASPP_ResNet.py:289)
  ubresnet_amd.training.pixelwise_nllloss.PixelWiseNLLLoss (training/pixelwise_nllloss.py:34)
All tensor arithmetic runs in hand-written HIP kernels (ubresnet_amd/csrc) behind the C ABI
declared in include/ubresnet_hip.h; there is no CPU or eager-PyTorch fallback.
"""
__version__ = "0.1.0"
