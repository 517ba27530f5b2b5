"""Block-level parity (GPU): BasicBlock / ConvTransposeLayer through the graph executor against the
reference-generated fixture tests/golden/blocks.npz (forward) and the CPU oracle's autograd
(gradients).  fp32: forward <= 1e-4 of scale, gradients <= 5e-4 of scale."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from oracle import uresnet_oracle as O

if torch.cuda.is_available():
    from ubresnet_amd.engine import Engine, Saved
    from ubresnet_amd.models.common_layers import BasicBlock, ConvTransposeLayer

DEV = "cuda"


class _Wrap(nn.Module):
    def __init__(self, mod):
        super().__init__()
        self.mod = mod

    def _grad_completion_order(self):
        return self.mod._grad_completion_order("mod.")


def nhwc(t):
    return torch.from_numpy(np.ascontiguousarray(t)).permute(0, 2, 3, 1).contiguous().to(DEV)


def nchw(t):
    return t.float().permute(0, 3, 1, 2).cpu()


def _rel(a, b):
    return float((a - b).abs().max() / max(b.abs().max().item(), 1e-9))


def _run(eng, fwd, bwd, go):
    sv = Saved()
    eng._alloc_pass_workspaces(sv, torch.device(DEV), True)
    eng._save = True
    eng.pack_all(torch.float32, torch.device(DEV, 0), "fwd")
    eng.pack_all(torch.float32, torch.device(DEV, 0), "bwd")
    rec = fwd()
    flat = torch.zeros(eng.grad_numel, device=DEV)
    views = {}
    for name, p in eng.grad_order:
        o = eng.grad_offsets[name]
        views[id(p)] = flat[o:o + p.numel()].view(p.shape)
    gx = bwd(rec, lambda p: views[id(p)])
    eng._wg_flush()        # (a whole backward pass sums the weight-gradient slabs once per stage: Engine._stage_notifier)
    torch.cuda.synchronize()
    return gx, {name: views[id(p)].cpu() for name, p in eng.grad_order}


@pytest.mark.parametrize("name,cfg", [("id", (8, 8, 1)), ("proj", (8, 16, 1)), ("down", (8, 16, 2))])
def test_basic_block(golden_dir, name, cfg):
    cin, cout, stride = cfg
    g = np.load(os.path.join(golden_dir, "blocks.npz"))
    # fixtures use 8/16 channels; the MFMA path needs multiples of 16 -> embed in zero-padded channels
    sd = O.seeded_state_dict(O._block_keys("b", cin, cout, stride), 50 + stride + cout)
    P = 16
    blk = BasicBlock(P, P, stride) if cin == cout else BasicBlock(P, P * (cout // cin), stride)
    co_p = blk.conv1.out_channels
    with torch.no_grad():
        for k, v in blk.state_dict().items():
            src = sd["b." + k]
            if v.dim() == 0:
                continue
            v.zero_()
            if k.endswith("running_var") or (k.endswith(".weight") and v.dim() == 1):
                v.fill_(1.0)
            sl = tuple(slice(0, s) for s in src.shape)
            v[sl] = src
    blk = blk.to(DEV).train()
    x = g["block_%s_x" % name]
    xp = np.zeros((x.shape[0], P, x.shape[2], x.shape[3]), np.float32)
    xp[:, :cin] = x
    eng = Engine(_Wrap(blk), "custom")
    xd = nhwc(xp)
    OH, OW = x.shape[2] // stride, x.shape[3] // stride
    out = torch.empty((x.shape[0], OH, OW, co_p), device=DEV)
    go = torch.randn(x.shape[0], cout, OH, OW, generator=torch.Generator().manual_seed(3))
    gop = torch.zeros(x.shape[0], co_p, OH, OW)
    gop[:, :cout] = go
    gx, grads = _run(eng, lambda: eng.block_fwd(blk, xd, out, True, torch.float32),
                     lambda rec, G: eng.block_bwd(rec, nhwc(gop.numpy()), None, G), None)
    ref = torch.from_numpy(g["block_%s_train" % name])
    assert _rel(nchw(out)[:, :cout], ref) <= 1e-4, "block forward vs reference fixture"
    assert nchw(out)[:, cout:].abs().max() <= 1e-5
    # gradients vs oracle autograd
    p = OrderedDict((k, v.clone().requires_grad_(True) if O.is_param_key(k) else v) for k, v in sd.items())
    xr = torch.from_numpy(x).requires_grad_(True)
    y = O.basic_block(p, "b", xr, stride, True, None)
    y.backward(go)
    assert _rel(nchw(gx)[:, :cin], xr.grad) <= 5e-4, "block input gradient"
    for k, v in p.items():
        if not O.is_param_key(k):
            continue
        got = grads["mod." + k[2:]]
        sl = tuple(slice(0, s) for s in v.shape)
        assert _rel(got[sl], v.grad) <= 5e-4, "gradient of %s: %.3e" % (k, _rel(got[sl], v.grad))


def test_conv_transpose_layer(golden_dir):
    g = np.load(os.path.join(golden_dir, "blocks.npz"))
    sd = O.seeded_state_dict(OrderedDict([("d.deconv.weight", (16, 8, 4, 4))] + list(O._double_keys("d.res", 16, 8, 1).items())), 60)
    # fixture: deconv 16->8, skip 8, res 16->8.  Padded to 16->16, skip 16, res 32->16 with zero channels.
    ctl = ConvTransposeLayer(16, 16, 16)
    with torch.no_grad():
        for k, v in ctl.state_dict().items():
            src = sd["d." + k]
            if v.dim() == 0:
                continue
            v.zero_()
            if k.endswith("running_var") or (k.endswith(".weight") and v.dim() == 1):
                v.fill_(1.0)
            if k in ("res.res1.conv1.weight", "res.res1.bypass.weight"):
                # input channels: [up 0..7 | pad 8..15 | skip 16..23 | pad 24..31]
                v[:8, 0:8] = src[:, 0:8]
                v[:8, 16:24] = src[:, 8:16]
            else:
                sl = tuple(slice(0, s) for s in src.shape)
                v[sl] = src
    ctl = ctl.to(DEV).train()
    x, skip = g["ctl_x"], g["ctl_skip"]
    N, _, H, W = x.shape
    eng = Engine(_Wrap(ctl), "custom")
    cat = torch.zeros((N, 2 * H, 2 * W, 32), device=DEV)
    cat[..., 16:24] = nhwc(skip)
    out = torch.empty((N, 2 * H, 2 * W, 16), device=DEV)
    go = torch.randn(N, 8, 2 * H, 2 * W, generator=torch.Generator().manual_seed(4))
    gop = torch.zeros(N, 16, 2 * H, 2 * W)
    gop[:, :8] = go
    xd = nhwc(x)
    res, grads = _run(eng, lambda: eng.declayer_fwd(ctl, xd, cat, 16, out, True, torch.float32),
                      lambda rec, G: eng.declayer_bwd(rec, nhwc(gop.numpy()), G), None)
    gx, g_cat = res
    assert _rel(nchw(out)[:, :8], torch.from_numpy(g["ctl_train"])) <= 1e-4, "ConvTransposeLayer forward vs reference fixture"
    p = OrderedDict((k, v.clone().requires_grad_(True) if O.is_param_key(k) else v) for k, v in sd.items())
    xr, sr = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(skip).requires_grad_(True)
    y = O.conv_transpose_layer(p, "d", xr, sr, True, None)
    y.backward(go)
    assert _rel(nchw(gx), xr.grad) <= 5e-4, "deconv input gradient"
    assert _rel(nchw(g_cat)[:, 16:24], sr.grad) <= 5e-4, "skip gradient"
    got = grads["mod.deconv.weight"][:, :8]
    assert _rel(got, p["d.deconv.weight"].grad) <= 5e-4, "deconv weight gradient"
    got = grads["mod.res.res2.conv2.weight"][:8, :8]
    assert _rel(got, p["d.res.res2.conv2.weight"].grad) <= 5e-4


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(2, 1, 40, 64, 16), (1, 3, 32, 32, 32)])
def test_stem_on_matrix_cores(cfg, dt):
    """conv1 7x7 (+bias, BN statistics) and its weight/bias gradients through the 16-channel column expansion."""
    import torch.nn.functional as F
    from ubresnet_amd import nn_params as P
    N, Cin, H, W, Cout = cfg

    class Stem(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = P.Conv2d(Cin, Cout, kernel_size=7, stride=1, padding=3, bias=True)

        def _grad_completion_order(self, prefix):
            return [(prefix + "conv1.weight", self.conv1.weight), (prefix + "conv1.bias", self.conv1.bias)]

    st = Stem().to(DEV)
    eng = Engine(_Wrap(st), "custom")
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, Cin, H, W, generator=g)
    x[x.abs() < 1.0] = 0.0
    xq = x.to(dt).float()                       # the expansion stores the image in the compute type
    w, b = st.conv1.weight.detach().cpu(), st.conv1.bias.detach().cpu()
    wr, br = w.to(dt).float().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv2d(xq, wr, br, 1, 3)
    c0 = torch.empty((N, H, W, Cout), dtype=dt, device=DEV)
    stats = torch.zeros(32 * 2 * Cout, dtype=torch.float64, device=DEV)
    eng.pack_all(dt, torch.device(DEV, 0), "fwd")
    x16 = eng.stem_fwd(st.conv1, x.to(DEV), c0, stats, dt)
    torch.cuda.synchronize()
    tol = 2e-5 if dt == torch.float32 else 1.2e-2
    assert _rel(nchw(c0), ref.detach()) <= tol
    assert _rel(stats.view(32, 2 * Cout).sum(0)[:Cout].cpu().float(), ref.detach().sum(dim=(0, 2, 3))) <= 10 * tol
    go = torch.randn(N, Cout, H, W, generator=g).to(dt).float()
    ref.backward(go)
    flat = torch.zeros(eng.grad_numel, device=DEV)
    views = {}
    for name, p in eng.grad_order:
        o = eng.grad_offsets[name]
        views[id(p)] = flat[o:o + p.numel()].view(p.shape)
    eng.stem_bwd(st.conv1, x16, go.permute(0, 2, 3, 1).contiguous().to(dt).to(DEV), lambda p: views[id(p)])
    eng._wg_flush()
    torch.cuda.synchronize()
    assert _rel(views[id(st.conv1.weight)].cpu(), wr.grad) <= 4 * tol
    assert _rel(views[id(st.conv1.bias)].cpu(), br.grad) <= 4 * tol
