"""Mirror of the reference's ``training`` package for the pieces on the hot path
(training/pixelwise_nllloss.py; metrics of training/train_ubresnet2018_wlarcv2.py:509-566)."""
