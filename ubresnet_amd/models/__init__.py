"""Mirror of the reference's ``models`` package (models/__init__.py): importable both as
``ubresnet_amd.models.ub_uresnet`` and, with this directory on sys.path (UBRESNET_MODELDIR,
training/train_ubresnet2018_wlarcv2.py:37-41), as top-level ``ub_uresnet`` / ``common_layers`` /
``ASPP_ResNet``."""
