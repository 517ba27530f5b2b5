"""Inference path (GPU): checkpoint loading (deploy/ubresnet_funcs.py:41-68 contract), pre-cropped
eval forward vs the reference fixture, whole-view tiling: crop/stitch bit-exact against torch slicing,
hipGraph replay bit-exact against eager launches."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import uresnet_oracle as O
from ubresnet_amd import synthetic

if torch.cuda.is_available():
    from ubresnet_amd import deploy


def test_load_model_and_precropped(golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "uresnet_ip16_nc4_1x1x64x96.npz"))
    B, C, H, W, seed0, wseed = [int(v) for v in g["meta"]]
    sd = O.seeded_state_dict(O.uresnet_schema(4, C, 16, 16), wseed)
    # a checkpoint as the reference writes it from a DataParallel model (training/train_ubresnet2018_wlarcv2.py:260-266)
    ck = {"iter": 7, "epoch": 0, "state_dict": {"module." + k: v for k, v in sd.items()}, "best_prec1": 0.0, "optimizer": {}}
    path = deploy.save_checkpoint(ck, False, -1, str(tmp_path / "checkpoint.pth.tar"))
    m = deploy.load_model(path, "cuda:0", num_classes=4)
    assert not m.training
    x, _, _ = synthetic.make_batch(B, H, W, seed0)
    out = deploy.segment_crops(m, torch.from_numpy(x).cuda(), batch=1).cpu()
    ref = torch.from_numpy(g["logp_eval"])
    assert float((out - ref).abs().max() / ref.abs().max()) <= 1e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_whole_view_tiling(dtype):
    sd = O.seeded_state_dict(O.uresnet_schema(3, 1, 16, 16), 42)
    m = deploy.load_model(None, "cuda:0", num_classes=3, state_dict=sd)
    P, rows, cols, th, tw = 2, 100, 200, 64, 96
    rs = np.random.RandomState(3)
    # random (untrained) weights with eval-mode running statistics do not normalise the activations, so keep the
    # fp16 case inside the fp16 range with a small input scale
    amp = 100.0 if dtype == torch.float32 else 0.5
    view = torch.from_numpy((rs.rand(P, 1, rows, cols) * (rs.rand(P, 1, rows, cols) > 0.9) * amp).astype(np.float32)).cuda()
    seg = deploy.WholeViewSegmenter(m, rows, cols, planes=P, tile=(th, tw), batch=4, dtype=dtype, use_graph=True)
    ro, co = deploy.regular_tiling(rows, cols, th, tw)
    assert ro == [0, 36] and co == [0, 52, 104]
    out = seg(view)
    assert torch.isfinite(out).all()
    out2 = seg(view)                      # graph replay is repeatable
    assert torch.equal(out, out2)
    eager = deploy.WholeViewSegmenter(m, rows, cols, planes=P, tile=(th, tw), batch=4, dtype=dtype, use_graph=False)
    assert torch.equal(out, eager(view)), "hipGraph replay differs from eager launches"
    # reference stitch in torch: every pixel from the tile that keeps it
    want = torch.full_like(out, float("nan"))
    m.compute_dtype = dtype
    with torch.no_grad():
        for (p, r0, c0, kr0, kr1, kc0, kc1) in seg.tiles:
            crop = torch.zeros((1, 1, th, tw), device="cuda")
            hh, ww = min(th, rows - r0), min(tw, cols - c0)
            crop[0, 0, :hh, :ww] = view[p, 0, r0:r0 + hh, c0:c0 + ww]
            sc = m(crop)[0]
            y1, x1 = min(kr1, rows - r0), min(kc1, cols - c0)
            want[p, :, r0 + kr0:r0 + y1, c0 + kc0:c0 + x1] = sc[:, kr0:y1, kc0:x1]
    m.compute_dtype = None
    assert not torch.isnan(want).any(), "keep windows must partition the view"
    assert torch.equal(out, want)


def test_full_size_tiling_geometry():
    ro, co = deploy.regular_tiling(1008, 3456, 512, 832)
    assert ro == [0, 496] and co == [0, 656, 1312, 1968, 2624]          # 10 tiles/plane, 30/event
    rk, ck = deploy._keep_windows(ro, 512, 1008), deploy._keep_windows(co, 832, 3456)
    assert rk[0][0] == 0 and rk[-1][1] == 1008 and all(a[1] == b[0] for a, b in zip(rk[:-1], rk[1:]))
    assert ck[0][0] == 0 and ck[-1][1] == 3456 and all(a[1] == b[0] for a, b in zip(ck[:-1], ck[1:]))
