"""Alias of ``ASPP_ResNet`` under the module name the reference's train scripts import
(``from ASPP_ResNet1 import ASPP_ResNet``, training/Sem_Seg_ASPP_ResNet1.py:43;
``from models.ASPP_ResNet1 import ASPP_ResNet``, training/grid_scripts/train_aspp_wlarcv1_tuftsgrid.py:38).
The reference tree only holds ``models/ASPP_ResNet.py``, so those imports fail there;
here both styles resolve to the one module object of ``ubresnet_amd.models.ASPP_ResNet``."""
import importlib as _il
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

sys.modules[__name__] = _il.import_module("ubresnet_amd.models.ASPP_ResNet")
