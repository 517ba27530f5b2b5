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


def _norm_model(g, device="cuda:0"):
    B, C, H, W, seed0, wseed = [int(v) for v in g["meta"]]
    sd = O.state_dict_with_bn_stats(O.seeded_state_dict(O.uresnet_schema(4, C, 16, 16), wseed), g["bn_keys"], g["bn_stats"])
    m = deploy.load_model(None, device, num_classes=4, state_dict=sd)
    x, _, _ = synthetic.make_batch(B, H, W, seed0)
    return m, torch.from_numpy(x).cuda()


# fp32: north_star's 1e-3 relative, per element (log-probabilities are O(10) here, so the relative form is meaningful).
# fp16 / bf16 storage with fp32 accumulation through ~60 layers, measured worst absolute errors on |logp| <= 63:
# fp32 2.7e-5 / 5.7e-5 (64x96 / 512x832), fp16 2.8e-2 / 6.6e-2, bf16 1.7e-1.  Bars = (relative, absolute) per element.
_EVAL_TOL = {torch.float32: (1e-3, 1e-4), torch.float16: (1e-2, 8e-2), torch.bfloat16: (5e-2, 4e-1)}


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_eval_forward_matches_normalised_reference_fixture(golden_dir, dtype):
    """the inference schedule (BatchNorm folded into packed weights, ReLU / shortcut in the conv epilogue) against the
    REFERENCE's eval forward on a deployment-like state (running statistics calibrated by the reference itself)"""
    g = np.load(os.path.join(golden_dir, "uresnet_ip16_nc4_norm_1x1x64x96.npz"))
    m, x = _norm_model(g)
    m.compute_dtype = dtype
    with torch.no_grad():
        out = m(x).cpu()
    ref = torch.from_numpy(g["logp_eval"])
    rel, atol = _EVAL_TOL[dtype]
    d = (out - ref).abs()
    print("eval %s: max abs err %.3e (|logp| max %.2f)" % (dtype, float(d.max()), float(ref.abs().max())))
    assert bool((d <= rel * ref.abs() + atol).all()), "worst excess %.3e" % float((d - rel * ref.abs()).max())
    top2 = torch.topk(ref, 2, dim=1)[0]
    safe = (top2[:, 0] - top2[:, 1]) > 4 * atol
    assert torch.equal(out.argmax(1)[safe], ref.argmax(1)[safe])
    assert abs(float(torch.logsumexp(out, 1).abs().max())) <= 1e-3          # rows are log-probabilities


def test_inference_schedule_equals_training_schedule_in_eval_mode(golden_dir, monkeypatch):
    """folding is an exact re-association in exact arithmetic: the fp32 results of the two schedules agree to rounding"""
    from ubresnet_amd import engine
    g = np.load(os.path.join(golden_dir, "uresnet_ip16_nc4_norm_1x1x64x96.npz"))
    m, x = _norm_model(g)
    with torch.no_grad():
        folded = m(x)
        monkeypatch.setattr(engine, "_INFER_FOLD", False)
        plain = m(x)
    assert float((folded - plain).abs().max()) <= 2e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_full_size_tile_matches_reference_summary(golden_dir, dtype):
    """one 512x832 tile (deploy/run_ubresnet_wholeview.py:38-39) at its real size: sampled log-probabilities and the
    class map of the reference's eval forward"""
    import hashlib
    g = np.load(os.path.join(golden_dir, "uresnet_ip16_nc4_norm_1x1x512x832_summary.npz"))
    m, x = _norm_model(g)
    m.compute_dtype = dtype
    with torch.no_grad():
        out = m(x).cpu().numpy()
    rel, atol = _EVAL_TOL[dtype]
    ref = g["sample_logp_eval"]
    d = np.abs(out.reshape(-1)[g["sample_idx"]] - ref)
    print("512x832 %s: sample max abs err %.3e" % (dtype, float(d.max())))
    assert bool((d <= rel * np.abs(ref) + atol).all())
    am = out.argmax(1).astype(np.uint8).reshape(-1)
    ram = g["argmax"].reshape(-1)
    if dtype == torch.float32:
        safe = np.unpackbits(g["safe_0p02"])[:am.size].astype(bool)
        assert np.array_equal(am[safe], ram[safe])
        if int(g["margin_hist"][:2].sum()) == 0:
            assert hashlib.sha256(am.tobytes()).hexdigest() == str(g["argmax_sha256"])
    else:
        safe = np.unpackbits(g["safe_0p2"])[:am.size].astype(bool)
        assert float((am[safe] == ram[safe]).mean()) >= 0.9995
        cm = np.bincount(ram[safe].astype(np.int64) * 4 + am[safe], minlength=16).reshape(4, 4)
        iou = [cm[c, c] / max(1, cm[c].sum() + cm[:, c].sum() - cm[c, c]) for c in range(4)]
        print("fp16 IoU vs reference (margin > 0.2):", iou, "agreement everywhere %.5f" % float((am == ram).mean()))
        assert min(iou) >= 0.99


def test_whole_view_full_size_event(golden_dir):
    """BASELINE configs[4] at its real size: a 3 x 1008 x 3456 event = 30 tiles of 512x832 (10 per plane), fp16, three
    hipGraph replays of 10 tiles.  Graph replay == eager launches bitwise; keep windows partition the view; the stitched
    scores equal the per-tile eager forward in every keep window; the tile at the fixture's position reproduces the
    reference's class map."""
    g = np.load(os.path.join(golden_dir, "uresnet_ip16_nc4_norm_1x1x512x832_summary.npz"))
    m, x = _norm_model(g)
    P, rows, cols, th, tw = 3, 1008, 3456, 512, 832
    view = torch.zeros((P, 1, rows, cols), device="cuda")
    for p in range(P):
        view[p, 0] = torch.from_numpy(synthetic.make_crop(rows, cols, 5000 + p)[0]).cuda()
    view[0, 0, :th, :tw] = x[0, 0]                      # tile (plane 0, row 0, col 0) is the fixture's input
    seg = deploy.WholeViewSegmenter(m, rows, cols, planes=P, tile=(th, tw), batch=10, dtype=torch.float16, use_graph=True)
    assert seg.tiles_per_event == 30
    cover = torch.zeros((P, rows, cols), dtype=torch.int32)
    for (p, r0, c0, kr0, kr1, kc0, kc1) in seg.tiles:
        cover[p, r0 + kr0:min(r0 + kr1, rows), c0 + kc0:min(c0 + kc1, cols)] += 1
    assert int(cover.min()) == 1 and int(cover.max()) == 1, "keep windows must partition the view"
    out = seg(view)
    assert out.shape == (P, 4, rows, cols) and torch.isfinite(out).all()
    assert torch.equal(out, seg(view))
    eager = deploy.WholeViewSegmenter(m, rows, cols, planes=P, tile=(th, tw), batch=10, dtype=torch.float16, use_graph=False)
    assert torch.equal(out, eager(view)), "hipGraph replay differs from eager launches"
    whole = deploy.WholeViewSegmenter(m, rows, cols, planes=P, tile=(th, tw), batch=30, dtype=torch.float16, use_graph=True)
    assert torch.equal(out, whole(view)), "one replay of 30 tiles (bench.py's setting) differs from three replays of 10"
    del whole
    m.compute_dtype = torch.float16
    with torch.no_grad():
        for i in (0, 7, 19, 29):                        # spot tiles: per-tile eager forward == the stitched keep window
            p, r0, c0, kr0, kr1, kc0, kc1 = seg.tiles[i]
            crop = torch.zeros((1, 1, th, tw), device="cuda")
            hh, ww = min(th, rows - r0), min(tw, cols - c0)
            crop[0, 0, :hh, :ww] = view[p, 0, r0:r0 + hh, c0:c0 + ww]
            sc = m(crop)[0]
            y1, x1 = min(kr1, rows - r0), min(kc1, cols - c0)
            assert torch.equal(out[p, :, r0 + kr0:r0 + y1, c0 + kc0:c0 + x1], sc[:, kr0:y1, kc0:x1]), "tile %d" % i
    m.compute_dtype = None
    # the fixture tile: the reference's class map where its top-2 margin is comfortable for fp16
    p, r0, c0, kr0, kr1, kc0, kc1 = seg.tiles[0]
    assert (p, r0, c0) == (0, 0, 0)
    am = out[0, :, kr0:kr1, kc0:kc1].argmax(0).cpu().numpy()
    ram = g["argmax"].reshape(th, tw)[kr0:kr1, kc0:kc1]
    safe = np.unpackbits(g["safe_0p2"])[:th * tw].astype(bool).reshape(th, tw)[kr0:kr1, kc0:kc1]
    assert float((am[safe] == ram[safe]).mean()) >= 0.9995
