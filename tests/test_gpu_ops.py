"""Per-kernel parity tests (GPU): every C-ABI entry point against a plain PyTorch fp32 CPU
restatement of the same op, on seeded inputs.  Integer outputs bit-exact; fp32 within 2e-5 of
the output scale (MFMA fp32 is an exact fmaf chain; only summation order differs); bf16/f16
within 1.2e-2 of the output scale with both sides fed the SAME rounded operands.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from ubresnet_amd import ops
    from ubresnet_amd.ops import Affine

DEV = "cuda"
DTS = [torch.float32, torch.bfloat16]


def tol(dt):
    return 2e-5 if dt == torch.float32 else 1.2e-2


def rnd(dt, t):
    """round to the compute type and come back (so both sides see identical operands)"""
    return t.to(dt).float()


def nhwc(t, dt):
    return t.permute(0, 2, 3, 1).contiguous().to(dt).to(DEV)


def nchw(t):
    return t.float().permute(0, 3, 1, 2).cpu()


def close(got, ref, rel, what=""):
    scale = max(ref.abs().max().item(), 1e-6)
    err = (got - ref).abs().max().item()
    assert err <= rel * scale, "%s: max err %.3e vs scale %.3e (rel %.2e > %.2e)" % (what, err, scale, err / scale, rel)


def gen(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def lo_zero(n):
    return torch.zeros(n, device=DEV)


SLOTS = 32   # UBR_STAT_SLOTS: fp64 accumulators are [slot][n]


def statbuf(n):
    return torch.zeros(SLOTS * n, dtype=torch.float64, device=DEV)


def slotsum(t, n):
    return t.view(SLOTS, n).sum(0)


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, dil, xf, bias, stats, addend, tile_hint
    (2, 24, 40, 16, 16, 3, 1, 1, True, False, True, False, 0),
    (1, 16, 32, 16, 16, 3, 1, 1, False, False, False, False, 1),
    (1, 16, 32, 16, 16, 3, 1, 1, False, True, True, True, 2),
    (2, 16, 16, 32, 64, 3, 1, 1, True, False, True, False, 0),
    (1, 16, 32, 32, 32, 3, 1, 1, False, False, True, False, 3),
    (1, 16, 32, 64, 64, 3, 1, 1, True, False, True, False, 4),
    (1, 8, 16, 64, 64, 3, 1, 1, False, False, False, False, 5),
    (1, 8, 16, 64, 128, 3, 1, 1, False, False, True, False, 6),
    (1, 8, 16, 32, 32, 3, 1, 1, False, False, False, False, 7),
    (1, 8, 16, 32, 16, 3, 1, 1, False, False, False, False, 8),
    (1, 8, 16, 32, 32, 3, 1, 1, False, False, False, False, 9),
    (1, 8, 16, 32, 16, 3, 1, 1, False, False, True, False, 10),
    (2, 32, 32, 16, 32, 3, 2, 1, False, False, True, False, 0),
    (1, 32, 64, 32, 64, 3, 2, 1, True, False, True, False, 0),
    (2, 16, 32, 16, 32, 1, 1, 1, False, False, True, False, 0),
    (2, 32, 32, 32, 64, 1, 2, 1, False, False, True, False, 0),
    (1, 32, 32, 16, 16, 7, 1, 1, False, True, True, False, 0),
    (1, 16, 24, 128, 16, 3, 1, 3, False, True, True, False, 0),
    (1, 16, 24, 64, 16, 3, 1, 5, False, True, False, False, 0),
    (1, 4, 8, 512, 64, 3, 1, 1, False, False, True, False, 0),
    (1, 12, 20, 48, 32, 3, 1, 1, True, False, True, False, 0),   # ragged: not tile multiples, Cin=48
    (1, 96, 128, 16, 16, 7, 1, 1, True, True, True, False, 0),    # head shapes of the 1x1x96x128 fixture
    (1, 96, 128, 16, 16, 3, 1, 1, False, False, True, True, 0),
    (1, 48, 64, 32, 32, 3, 1, 1, True, False, True, False, 0),
    (2, 64, 96, 16, 16, 7, 1, 1, False, True, False, False, 0),
    # 7x7 over 16 channels: tiles of 16x32 take the row-stationary thin kernel (ROW7) in 16-bit types; hint 2 = the generic tap loop
    (1, 32, 64, 16, 16, 7, 1, 1, True, True, True, True, 0),
    (1, 32, 64, 16, 16, 7, 1, 1, True, True, True, True, 2),
    (2, 16, 32, 16, 12, 7, 1, 1, False, False, True, False, 0),
    # conv_pc_kernel (producer / consumer waves, persistent over (cout tile, pixel tile) units): tiles 101..104 = 8x32, 4x32, 16x16, 8x16 px
    (1, 16, 32, 64, 128, 3, 1, 1, True, False, True, False, 101),
    (1, 8, 32, 128, 128, 3, 1, 1, False, True, True, True, 102),
    (1, 32, 32, 64, 64, 3, 1, 1, True, False, True, False, 103),
    (2, 16, 16, 128, 128, 3, 1, 1, False, False, True, True, 104),
    (1, 8, 16, 64, 128, 3, 1, 1, True, True, False, False, 104),
    (1, 16, 16, 32, 64, 3, 1, 1, False, True, True, False, 103),      # a single cin block
    (1, 12, 40, 64, 128, 3, 1, 1, True, False, True, False, 101),    # ragged edges
    (2, 20, 24, 96, 192, 3, 1, 1, False, False, True, False, 0),      # auto choice: three cout tiles, 16-pixel-wide tiles
    (1, 16, 32, 64, 128, 1, 1, 1, False, False, True, False, 101),    # 1x1: one K-step per block
    (3, 128, 128, 64, 128, 3, 1, 1, True, True, True, True, 0),       # 384 units on <= 256 workgroups: units of both cout tiles per workgroup
    (2, 32, 64, 256, 64, 3, 1, 1, True, False, True, False, 0),       # eight cin blocks
]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward(case, dt):
    N, H, W, Cin, Cout, k, stride, dil, xf, bias, stats, addend, hint = case
    pad = dil * (k // 2)
    x = rnd(dt, gen(N, Cin, H, W, seed=1))
    w = gen(Cout, Cin, k, k, seed=2, scale=(2.0 / (k * k * Cin)) ** 0.5)
    b = gen(Cout, seed=3) if bias else None
    xin = x
    aff = None
    if xf:
        sc, sh = gen(Cin, seed=4).abs() + 0.5, gen(Cin, seed=5) * 0.3
        xin = rnd(dt, F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)))
        aff = Affine(lo_zero(Cin), sc.to(DEV), sh.to(DEV), lo_zero(Cin))
    ref = F.conv2d(xin, rnd(dt, w), b, stride, pad, dil)
    OH, OW = ref.shape[2], ref.shape[3]
    ad = None
    if addend:
        ad = rnd(dt, gen(N, Cout, OH, OW, seed=6))
        ref = ref + ad
    # input as a channel slice of a wider buffer, output as a channel slice too
    xbuf = torch.zeros((N, H, W, Cin + 16), dtype=dt, device=DEV)
    xbuf[..., 16:] = nhwc(x, dt)
    ybuf = torch.full((N, OH, OW, Cout + 32), 7.0, dtype=dt, device=DEV)
    yv = ybuf[..., 32:]
    wp = ops.pack_weights(w.to(DEV), dt, Cout, Cin, Cin * k * k, k * k, k * k)
    st = statbuf(2 * Cout) if stats else None
    ops.conv(xbuf[..., 16:], wp, yv, ops.conv_taps(k, dil, pad), Cout, S=stride, xf=aff,
             bias=b.to(DEV) if bias else None, addend=nhwc(ad, dt) if addend else None, stats=st, tile_hint=hint)
    torch.cuda.synchronize()
    close(nchw(yv), ref, tol(dt), "conv out")
    assert (ybuf[..., :32].float() == 7.0).all(), "conv wrote outside its channel slice"
    if stats:
        s = slotsum(st, 2 * Cout).cpu()
        n = N * OH * OW
        rs, rss = ref.sum(dim=(0, 2, 3)).double(), (ref.double() ** 2).sum(dim=(0, 2, 3))
        assert (s[:Cout] - rs).abs().max().item() <= tol(dt) * n * max(ref.abs().max().item(), 1e-6)
        assert (s[Cout:] - rss).abs().max().item() <= 2 * tol(dt) * rss.max().item() + 1e-6


@pytest.mark.parametrize("dt", DTS + [torch.float16])
@pytest.mark.parametrize("shape", [(2, 32, 64, 128, 128, 101, 4), (1, 16, 16, 256, 64, 104, 5), (3, 64, 64, 64, 128, 102, 4)])
def test_conv_pc_equals_igemm_bitwise(dt, shape):
    """conv_pc_kernel keeps conv_igemm_kernel's order of operations (cin blocks, taps, units): bitwise the same output, whichever
    kernel or tile a launch gets; BatchNorm-on-load, bias, addend and both activations in the epilogue."""
    N, H, W, Cin, Cout, hint_pc, hint_ig = shape
    x = rnd(dt, gen(N, Cin, H, W, seed=11))
    w = gen(Cout, Cin, 3, 3, seed=12, scale=(2.0 / (9 * Cin)) ** 0.5)
    b = gen(Cout, seed=13).to(DEV)
    sc, sh = gen(Cin, seed=14).abs() + 0.5, gen(Cin, seed=15) * 0.3
    aff = Affine(lo_zero(Cin), sc.to(DEV), sh.to(DEV), lo_zero(Cin))
    ad = nhwc(rnd(dt, gen(N, Cout, H, W, seed=16)), dt)
    wp = ops.pack_weights(w.to(DEV), dt, Cout, Cin, Cin * 9, 9, 9)
    xd = nhwc(x, dt)
    outs, sts = [], []
    for hint in (hint_pc, hint_ig):
        y = torch.empty((N, H, W, Cout), dtype=dt, device=DEV)
        st = statbuf(2 * Cout)
        ops.conv(xd, wp, y, ops.conv_taps(3, 1, 1), Cout, xf=aff, bias=b, addend=ad, stats=st, act=3, tile_hint=hint)
        torch.cuda.synchronize()
        outs.append(y)
        sts.append(slotsum(st, 2 * Cout))
    assert ops.last_conv_kernel().startswith("conv_igemm_kernel")
    assert torch.equal(outs[0], outs[1])
    assert torch.allclose(sts[0], sts[1], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("dt", DTS + [torch.float16])
@pytest.mark.parametrize("shape", [(2, 32, 64, 16, 16, 3), (1, 16, 64, 32, 32, 3), (2, 16, 32, 64, 64, 3), (1, 8, 16, 128, 128, 3), (1, 32, 64, 16, 16, 7),
                                   (1, 12, 20, 48, 32, 3)])
def test_conv_training_epilogues(dt, shape):
    """ubr_conv_desc.bnb_c: the conv's `stats` are the BatchNorm-backward sums of a = relu(bn(c)) for g = its (stored) output --
    equal to a separate ubr_bn_bwd_reduce over the stored tensor; ubr_conv_desc.addend_mask: out = conv + addend * bit --
    bitwise what the unmasked conv gives on a pre-masked addend.  Thin (<= 32 channels), row-stationary 7x7 and generic kernels."""
    N, H, W, Cin, Cout, ksz = shape
    x = nhwc(rnd(dt, gen(N, Cin, H, W, seed=21)), dt)
    w = gen(Cout, Cin, ksz, ksz, seed=22, scale=(2.0 / (ksz * ksz * Cin)) ** 0.5)
    wp = ops.pack_weights(w.to(DEV), dt, Cout, Cin, Cin * ksz * ksz, ksz * ksz, ksz * ksz)
    taps = ops.conv_taps(ksz, 1, ksz // 2)
    c = nhwc(rnd(dt, gen(N, Cout, H, W, seed=23)), dt)
    mean, scale, shift, invstd = [(gen(Cout, seed=24 + i) * 0.3 + (0.0 if i in (0, 2) else 1.0)).to(DEV) for i in range(4)]
    y0 = torch.empty((N, H, W, Cout), dtype=dt, device=DEV)
    ops.conv(x, wp, y0, taps, Cout)
    red0 = statbuf(2 * Cout)
    ops.bn_bwd_reduce(y0, None, c, scale, shift, mean, invstd, True, red0)
    y1 = torch.full_like(y0, float("nan"))
    red1 = statbuf(2 * Cout)
    ops.conv(x, wp, y1, taps, Cout, bnb=(c, mean, scale, shift, invstd), stats=red1)
    torch.cuda.synchronize()
    assert torch.equal(y1, y0)
    a, b = slotsum(red1, 2 * Cout), slotsum(red0, 2 * Cout)
    assert torch.allclose(a, b, rtol=1e-5, atol=2e-6 * float(b.abs().max()) + 2e-5), "fused BatchNorm-backward sums: %.3e" % float((a - b).abs().max())
    assert float(red1.view(SLOTS, -1)[8:].abs().max()) == 0.0          # UBR_RED_SLOTS stripes only
    # masked addend
    cpu_ = 4 if dt == torch.float32 else 8
    if Cout % cpu_ == 0:
        ad = nhwc(rnd(dt, gen(N, Cout, H, W, seed=29)), dt)
        mask = torch.randint(0, 256, (N * H * W * (Cout // cpu_),), dtype=torch.uint8, device=DEV)
        bits = ((mask.view(N, H, W, Cout // cpu_, 1) >> torch.arange(cpu_, device=DEV, dtype=torch.uint8)) & 1).reshape(N, H, W, Cout).bool()
        ad_masked = torch.where(bits, ad, torch.zeros_like(ad))
        y2, y3 = torch.empty_like(y0), torch.full_like(y0, float("nan"))
        ops.conv(x, wp, y2, taps, Cout, addend=ad_masked)
        ops.conv(x, wp, y3, taps, Cout, addend=ad, addend_mask=mask)
        torch.cuda.synchronize()
        assert torch.equal(y3, y2)


@pytest.mark.parametrize("dt", DTS)
def test_wgrad_slab_sums_batched_equal_single_launches(dt):
    """ubr_wgrad_reduce_batched: the slab sums of several weight gradients in one launch (ops.ReduceBatch, as the executor issues
    them once per backward stage) are bitwise the sums of one ubr_wgrad_reduce launch each -- few slabs (the one-thread-per-element
    form), many slabs (four-wave form), a tap subset scattered into a larger kernel window, more than 16 items (two launches)."""
    ws = ops.WgradWorkspace()
    batch = ops.ReduceBatch(ws)
    cases = [(2, 32, 32, 64, 64, 3), (1, 64, 64, 16, 16, 3), (2, 16, 16, 128, 64, 1), (1, 48, 32, 32, 16, 3), (2, 8, 8, 256, 256, 3)]
    cases = cases * 4                        # 20 items: more than UBR_REDUCE_BATCH
    singles, batched, keep = [], [], []
    for i, (N, H, W, Cin, Cout, k) in enumerate(cases):
        x = nhwc(rnd(dt, gen(N, Cin, H, W, seed=100 + i)), dt)
        g = nhwc(rnd(dt, gen(N, Cout, H, W, seed=200 + i)), dt)
        taps = ops.conv_taps(k, 1, k // 2)
        if i % 5 == 3:
            taps = taps[::2]                 # a subset of the window (as the phases of a transposed conv)
        a = torch.full((Cout, Cin, k, k), 7.0, device=DEV)
        b = torch.full((Cout, Cin, k, k), 7.0, device=DEV)
        ops.wgrad(x, g, taps, a, Cin * k * k, k * k, Cout, Cin, ops.WgradWorkspace())
        ops.wgrad(x, g, taps, b, Cin * k * k, k * k, Cout, Cin, ws, defer=batch)
        singles.append(a); batched.append(b); keep.append((x, g))
    assert len(batch.items) == len(cases)
    batch.flush()
    assert not batch.items
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(singles, batched)):
        assert torch.equal(a, b), "item %d" % i
    # an accumulating call flushes what is pending first and then sums on its own
    a = singles[0].clone()
    x, g = keep[0]
    N, H, W, Cin, Cout, k = cases[0]
    ops.wgrad(x, g, ops.conv_taps(k, 1, 1), a, Cin * k * k, k * k, Cout, Cin, ws, accumulate=True, defer=batch)
    torch.cuda.synchronize()
    assert not batch.items
    assert torch.equal(a, singles[0] + singles[0])


@pytest.mark.parametrize("dt", DTS + [torch.float16])
@pytest.mark.parametrize("shape", [(2, 16, 16, 128, 64), (1, 8, 24, 512, 256), (2, 12, 20, 64, 64), (2, 32, 64, 32, 16), (1, 16, 32, 32, 32), (1, 32, 32, 16, 16)])
def test_conv_phases_equal_one_launch_per_phase(dt, shape):
    """ubr_conv_desc.nphase: the four output phases of ConvTranspose2d(k4, s2, p1) -- and of the data gradient of a stride-2
    3x3 conv, with an addend -- in ONE launch are bitwise the four launches (same kernel family, same cin-block order)."""
    N, H, W, Cin, Cout = shape
    x = nhwc(rnd(dt, gen(N, Cin, H, W, seed=31)), dt)
    aff = Affine(lo_zero(Cin), (gen(Cin, seed=34).abs() + 0.5).to(DEV), (gen(Cin, seed=35) * 0.3).to(DEV), lo_zero(Cin))
    for k, pad, seed, with_ad, xf in ((4, 1, 32, False, aff), (3, 1, 33, True, None)):
        w = gen(Cout, Cin, k, k, seed=seed, scale=(2.0 / (k * k * Cin)) ** 0.5)
        wp = ops.pack_weights(w.to(DEV), dt, Cout, Cin, Cin * k * k, k * k, k * k)
        ad = nhwc(rnd(dt, gen(N, Cout, 2 * H, 2 * W, seed=36)), dt) if with_ad else None
        ybuf = torch.full((N, 2 * H, 2 * W, Cout + 16), 3.0, dtype=dt, device=DEV)       # output as a channel slice
        ya = ybuf[..., :Cout]
        phases = [(ry, rx, ops.transposed_phase_taps(k, 1, pad, 2, ry, rx)) for ry in range(2) for rx in range(2)]
        assert all(p[2] for p in phases)
        for ry, rx, tp in phases:
            ops.conv(x, wp, ya[:, ry::2, rx::2, :], tp, Cout, xf=xf, addend=ad[:, ry::2, rx::2, :] if with_ad else None)
        assert ops.last_conv_kernel().startswith(("conv_igemm_kernel", "conv_pc_kernel", "conv_thin_kernel"))
        yb = torch.full((N, 2 * H, 2 * W, Cout), float("nan"), dtype=dt, device=DEV)
        ops.conv_phases(x, wp, yb[:, 0::2, 0::2, :], [t for p in phases for t in p[2]], Cout, phases=phases, y_full=yb, addend_full=ad, xf=xf)
        torch.cuda.synchronize()
        # (Cin <= 32 and Cout <= 32 without an addend: the thin kernel walks the phases over one staged halo)
        assert ops.last_conv_kernel().startswith(("conv_igemm_kernel", "conv_thin_kernel"))
        assert torch.equal(yb, ya.contiguous()), "k=%d %s" % (k, ops.last_conv_kernel())
        assert (ybuf[..., Cout:].float() == 3.0).all()


@pytest.mark.parametrize("dt", DTS + [torch.float16])
@pytest.mark.parametrize("chans", [(32, 16), (128, 64)])
def test_deconv_forward_and_grads(dt, chans):
    """ConvTranspose2d(k4,s2,p1) as 4 phases into a concat slice; its data and weight gradients."""
    N, H, W = 2, 8, 12
    Cin, Cd = chans
    x = rnd(dt, gen(N, Cin, H, W, seed=1))
    w = gen(Cin, Cd, 4, 4, seed=2, scale=0.1)
    xr = x.clone().requires_grad_(True)
    wr = rnd(dt, w).clone().requires_grad_(True)
    ref = F.conv_transpose2d(xr, wr, None, 2, 1)
    cat = torch.zeros((N, 2 * H, 2 * W, Cd + 16), dtype=dt, device=DEV)
    up = cat[..., :Cd]
    wp = ops.pack_weights(w.to(DEV), dt, Cd, Cin, 16, Cd * 16, 16)
    for ry in range(2):
        for rx in range(2):
            ops.conv(nhwc(x, dt), wp, up[:, ry::2, rx::2, :], ops.transposed_phase_taps(4, 1, 1, 2, ry, rx), Cd)
    torch.cuda.synchronize()
    close(nchw(up), ref.detach(), tol(dt), "deconv fwd")
    g = rnd(dt, gen(*ref.shape, seed=3))
    ref.backward(g)
    gcat = torch.zeros((N, 2 * H, 2 * W, Cd + 16), dtype=dt, device=DEV)
    gcat[..., :Cd] = nhwc(g, dt)
    gup = gcat[..., :Cd]
    # data gradient = stride-2 conv over g
    wpd = ops.pack_weights(w.to(DEV), dt, Cin, Cd, Cd * 16, 16, 16)
    gx = torch.empty((N, H, W, Cin), dtype=dt, device=DEV)
    ops.conv(gup, wpd, gx, ops.conv_taps(4, 1, 1), Cin, S=2)
    close(nchw(gx), xr.grad, tol(dt), "deconv dgrad")
    dW = torch.full((Cin, Cd, 4, 4), 5.0, device=DEV)
    ws = ops.WgradWorkspace()
    xd = nhwc(x, dt)
    for ry in range(2):
        for rx in range(2):
            ops.wgrad(xd, gup[:, ry::2, rx::2, :], ops.transposed_phase_taps(4, 1, 1, 2, ry, rx), dW, 16, Cd * 16, Cd, Cin, ws)
    torch.cuda.synchronize()
    close(dW.cpu(), wr.grad, 2 * tol(dt), "deconv wgrad")


GRAD_CASES = [
    # N, H, W, Cin, Cout, k, stride, xf
    (2, 24, 40, 16, 16, 3, 1, True),
    (1, 16, 32, 32, 32, 3, 1, False),
    (2, 16, 16, 64, 32, 3, 1, True),
    (2, 32, 32, 16, 32, 3, 2, False),
    (1, 16, 32, 32, 64, 1, 2, False),
    (2, 16, 16, 32, 16, 1, 1, False),
    (1, 24, 24, 16, 16, 7, 1, False),
    (1, 12, 20, 48, 32, 3, 1, False),
    (1, 4, 8, 256, 128, 3, 1, False),
    (1, 96, 128, 16, 16, 7, 1, True),
    (1, 96, 128, 16, 16, 3, 1, False),
    (1, 48, 64, 32, 32, 3, 1, True),
    (2, 64, 96, 32, 16, 3, 1, False),
    (1, 96, 128, 32, 16, 1, 1, False),
    (2, 16, 16, 64, 64, 3, 1, True),       # wide layers: N-split weight-gradient tiles (64 x 64 channels)
    (1, 16, 16, 64, 128, 3, 2, False),
    (1, 16, 16, 128, 64, 1, 1, False),
    (1, 8, 8, 128, 64, 1, 2, False),
    (2, 12, 20, 64, 64, 3, 1, False),
]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", GRAD_CASES)
def test_conv_dgrad_wgrad(case, dt):
    N, H, W, Cin, Cout, k, stride, xf = case
    pad = k // 2
    x = rnd(dt, gen(N, Cin, H, W, seed=1))
    w = gen(Cout, Cin, k, k, seed=2, scale=(2.0 / (k * k * Cin)) ** 0.5)
    aff, xin = None, x
    if xf:
        sc, sh = gen(Cin, seed=4).abs() + 0.5, gen(Cin, seed=5) * 0.3
        xin = rnd(dt, F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)))
        aff = Affine(lo_zero(Cin), sc.to(DEV), sh.to(DEV), lo_zero(Cin))
    xr = xin.clone().requires_grad_(True)
    wr = rnd(dt, w).clone().requires_grad_(True)
    y = F.conv2d(xr, wr, None, stride, pad)
    g = rnd(dt, gen(*y.shape, seed=3))
    y.backward(g)
    gd = nhwc(g, dt)
    # ---- weight gradient ----
    dW = torch.full((Cout, Cin, k, k), 3.0, device=DEV)
    ops.wgrad(nhwc(x, dt), gd, ops.conv_taps(k, 1, pad), dW, Cin * k * k, k * k, Cout, Cin, ops.WgradWorkspace(), S=stride, xf=aff)
    torch.cuda.synchronize()
    close(dW.cpu(), wr.grad, 2 * tol(dt), "wgrad")
    # accumulate flag
    ops.wgrad(nhwc(x, dt), gd, ops.conv_taps(k, 1, pad), dW, Cin * k * k, k * k, Cout, Cin, ops.WgradWorkspace(), S=stride, xf=aff, accumulate=True)
    close(dW.cpu(), 2 * wr.grad, 2 * tol(dt), "wgrad accumulate")
    # ---- data gradient (w.r.t. the conv input as the conv saw it) ----
    wpd = ops.pack_weights(w.to(DEV), dt, Cin, Cout, k * k, Cin * k * k, k * k)
    gx = torch.empty((N, H, W, Cin), dtype=dt, device=DEV)
    ad = rnd(dt, gen(N, Cin, H, W, seed=9))
    add = nhwc(ad, dt)
    if stride == 1:
        ops.conv(gd, wpd, gx, ops.conv_dgrad_taps_s1(k, 1, pad), Cin, addend=add)
    else:
        gx.copy_(add)
        for ry in range(2):
            for rx in range(2):
                taps = ops.transposed_phase_taps(k, 1, pad, 2, ry, rx)
                if taps:
                    ops.conv(gd, wpd, gx[:, ry::2, rx::2, :], taps, Cin, addend=gx[:, ry::2, rx::2, :])
    torch.cuda.synchronize()
    close(nchw(gx), xr.grad + ad, tol(dt), "dgrad")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("cfg", [(2, 1, 40, 56, 16), (1, 3, 32, 32, 32), (1, 1, 16, 16, 16)])
def test_stem(cfg, dt):
    N, Cin, H, W, Cout = cfg
    x = gen(N, Cin, H, W, seed=1)
    x[x.abs() < 1.2] = 0.0           # sparse like LArTPC crops
    x[0, :, :16, :16] = 0.0          # one all-zero tile -> exercises the skip path
    w = gen(Cout, Cin, 7, 7, seed=2, scale=0.2)
    b = gen(Cout, seed=3)
    ref = F.conv2d(x, w, b, 1, 3)
    y = torch.empty((N, H, W, Cout), dtype=dt, device=DEV)
    st = statbuf(2 * Cout)
    ops.stem_forward(x.to(DEV), w.to(DEV), b.to(DEV), y, st)
    torch.cuda.synchronize()
    close(nchw(y), ref, tol(dt), "stem fwd")
    ss = slotsum(st, 2 * Cout).cpu().float()
    close(ss[:Cout], ref.sum(dim=(0, 2, 3)), 1e-4, "stem stats sum")
    close(ss[Cout:], (ref ** 2).sum(dim=(0, 2, 3)), 1e-4, "stem stats sumsq")
    g = rnd(dt, gen(N, Cout, H, W, seed=4))
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    F.conv2d(x, wr, br, 1, 3).backward(g)
    dW = torch.empty_like(w, device=DEV)
    dB = torch.empty(Cout, device=DEV)
    ops.stem_wgrad(x.to(DEV), nhwc(g, dt), dW, dB, ops.WgradWorkspace())
    torch.cuda.synchronize()
    close(dW.cpu(), wr.grad, 1e-4, "stem wgrad")
    close(dB.cpu(), br.grad, 1e-4, "stem bgrad")


def _bn_vectors(C, seed):
    gamma = gen(C, seed=seed).abs() + 0.5
    beta = gen(C, seed=seed + 1) * 0.2
    return gamma, beta


@pytest.mark.parametrize("dt", DTS)
def test_bn_finalize(dt):
    C, n = 32, 5000
    v = gen(n, C, seed=1) * 2 + 3
    st = statbuf(2 * C)
    st[:2 * C] = torch.cat([v.double().sum(0), (v.double() ** 2).sum(0)]).to(DEV) * 0.25
    st[2 * C:4 * C] = st[:2 * C] * 3      # spread over two slots: the finalize sums all slots
    gamma, beta = _bn_vectors(C, 2)
    rm, rv = gen(C, seed=4).to(DEV), (gen(C, seed=5).abs() + 0.5).to(DEV)
    rm0, rv0 = rm.clone().cpu(), rv.clone().cpu()
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    out = [torch.empty(C, device=DEV) for _ in range(4)]
    ops.bn_finalize(st, n, gamma.to(DEV), beta.to(DEV), rm, rv, nbt, 0.1, 1e-5, *out)
    mean, var = v.mean(0), v.var(0, unbiased=False)
    invstd = 1 / torch.sqrt(var + 1e-5)
    close(out[0].cpu(), gamma * invstd, 1e-5, "scale")
    close(out[1].cpu(), beta, 1e-7, "shift")
    close(out[2].cpu(), mean, 1e-5, "mean")
    close(out[3].cpu(), invstd, 1e-5, "invstd")
    close(rm.cpu(), 0.9 * rm0 + 0.1 * mean, 1e-5, "running_mean")
    close(rv.cpu(), 0.9 * rv0 + 0.1 * v.var(0, unbiased=True), 1e-5, "running_var")
    assert int(nbt.item()) == 1
    sc, sh, mu, isd = [torch.empty(C, device=DEV) for _ in range(4)]
    ops.bn_eval_affine(gamma.to(DEV), beta.to(DEV), rm, rv, 1e-5, sc, sh, mu, isd)
    r_is = 1 / torch.sqrt(rv.cpu() + 1e-5)
    close(sc.cpu(), gamma * r_is, 1e-5, "eval scale")
    close(sh.cpu(), beta, 1e-7, "eval shift")
    close(mu.cpu(), rm.cpu(), 1e-7, "eval mean")
    close(isd.cpu(), r_is, 1e-5, "eval invstd")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("bypass", [False, True])
@pytest.mark.parametrize("C,N,H,W", [(16, 2, 12, 20), (96, 2, 12, 20), (16, 1, 96, 128), (512, 1, 6, 10), (64, 1, 40, 24)])
def test_block_tail(dt, bypass, C, N, H, W):
    """relu(relu(bn2(c2)) + shortcut) forward and the whole backward (both BatchNorms) vs autograd."""
    c2 = rnd(dt, gen(N, C, H, W, seed=1))
    sc_in = rnd(dt, gen(N, C, H, W, seed=2))
    g2, b2 = _bn_vectors(C, 3)
    gb, bb = _bn_vectors(C, 5)
    go1, go2 = rnd(dt, gen(N, C, H, W, seed=7)), rnd(dt, gen(N, C, H, W, seed=8))
    c2r, scr = c2.clone().requires_grad_(True), sc_in.clone().requires_grad_(True)
    g2r, b2r, gbr, bbr = [t.clone().requires_grad_(True) for t in (g2, b2, gb, bb)]
    r2 = F.relu(F.batch_norm(c2r, None, None, g2r, b2r, True, 0.1, 1e-5))
    sh = F.batch_norm(scr, None, None, gbr, bbr, True, 0.1, 1e-5) if bypass else scr
    out = F.relu(r2 + sh)
    out.backward(go1 + go2)

    def stats(t):
        m = t.mean(dim=(0, 2, 3))
        v = t.var(dim=(0, 2, 3), unbiased=False)
        return m, 1 / torch.sqrt(v + 1e-5)
    m2, i2 = stats(c2)
    mb, ib = stats(sc_in)
    s2, t2 = (g2 * i2), b2          # bn(x) = (x - mean)*scale + beta
    sb, tb = (gb * ib), bb
    d = lambda t: t.to(DEV)
    c2d, scd = nhwc(c2, dt), nhwc(sc_in, dt)
    outd = torch.empty((N, H, W, C), dtype=dt, device=DEV)
    ops.block_tail_fwd(c2d, d(m2), d(s2), d(t2), scd, d(mb) if bypass else None, d(sb) if bypass else None, d(tb) if bypass else None, outd)
    torch.cuda.synchronize()
    close(nchw(outd), out.detach(), tol(dt), "tail fwd")
    red2, redb = statbuf(2 * C), statbuf(2 * C)
    go1d, go2d = nhwc(go1, dt), nhwc(go2, dt)
    ops.block_tail_bwd_reduce(go1d, go2d, outd, c2d, d(s2), d(t2), d(m2), d(i2), scd if bypass else None,
                              d(mb) if bypass else None, d(ib) if bypass else None, red2, redb if bypass else None)
    k = torch.empty(4 * C, device=DEV)
    dg2, db2, dgb, dbb = [torch.empty(C, device=DEV) for _ in range(4)]
    cnt = N * H * W
    ops.bn_bwd_finalize(red2, cnt, C, dg2, db2, False, k[:C], k[C:2 * C])
    if bypass:
        ops.bn_bwd_finalize(redb, cnt, C, dgb, dbb, False, k[2 * C:3 * C], k[3 * C:])
    g_c2 = torch.empty((N, H, W, C), dtype=dt, device=DEV)
    g_sc = torch.empty((N, H, W, C), dtype=dt, device=DEV)
    ops.block_tail_bwd_apply(go1d, go2d, outd, c2d, d(s2), d(t2), d(m2), d(i2), k[:C], k[C:2 * C],
                             scd if bypass else None, d(sb) if bypass else None, d(mb) if bypass else None, d(ib) if bypass else None,
                             k[2 * C:3 * C] if bypass else None, k[3 * C:] if bypass else None, g_c2, g_sc)
    torch.cuda.synchronize()
    t = 4 * tol(dt)
    close(nchw(g_c2), c2r.grad, t, "g_c2")
    close(nchw(g_sc), scr.grad, t, "g_shortcut")
    close(dg2.cpu(), g2r.grad, t, "dgamma2")
    close(db2.cpu(), b2r.grad, t, "dbeta2")
    if bypass:
        close(dgb.cpu(), gbr.grad, t, "dgamma_b")
        close(dbb.cpu(), bbr.grad, t, "dbeta_b")
    # the masked form (what the executor runs): the forward also writes one bit per channel "stored output > 0", the backward
    # passes read those bytes instead of `out` -- bitwise the same results
    cpu_ = 4 if dt == torch.float32 else 8
    mask = torch.full((N * H * W * (C // cpu_) + 3,), 0xAA, dtype=torch.uint8, device=DEV)
    out_m = torch.empty_like(outd)
    ops.block_tail_fwd(c2d, d(m2), d(s2), d(t2), scd, d(mb) if bypass else None, d(sb) if bypass else None, d(tb) if bypass else None, out_m,
                       relu_mask=mask)
    torch.cuda.synchronize()
    assert torch.equal(out_m, outd)
    assert (mask[-3:] == 0xAA).all(), "mask written beyond npix * units"
    bits = (mask[:-3].view(N, H, W, C // cpu_, 1) >> torch.arange(cpu_, device=DEV, dtype=torch.uint8)) & 1
    assert torch.equal(bits.reshape(N, H, W, C).bool(), outd.float() > 0)
    red2m, redbm = statbuf(2 * C), statbuf(2 * C)
    junk = torch.full_like(outd, float("nan"))          # `out` must not be read in the masked form
    ops.block_tail_bwd_reduce(go1d, go2d, junk, c2d, d(s2), d(t2), d(m2), d(i2), scd if bypass else None,
                              d(mb) if bypass else None, d(ib) if bypass else None, red2m, redbm if bypass else None, relu_mask=mask)
    g_c2m, g_scm = torch.empty_like(g_c2), torch.empty_like(g_sc)
    ops.block_tail_bwd_apply(go1d, go2d, junk, c2d, d(s2), d(t2), d(m2), d(i2), k[:C], k[C:2 * C],
                             scd if bypass else None, d(sb) if bypass else None, d(mb) if bypass else None, d(ib) if bypass else None,
                             k[2 * C:3 * C] if bypass else None, k[3 * C:] if bypass else None, g_c2m, g_scm, relu_mask=mask)
    torch.cuda.synchronize()
    assert torch.equal(g_c2m, g_c2) and torch.equal(g_scm, g_sc)
    assert torch.allclose(slotsum(red2m, 2 * C), slotsum(red2, 2 * C), rtol=1e-12, atol=1e-9)
    # forward with the BatchNorm finalizes fused (ubr_block_tail_fwd_fin): from the striped conv statistics to the block output in
    # one launch -- bitwise what ubr_bn_finalize + ubr_block_tail_fwd_masked give, vectors, running statistics and counters included
    import torch.nn as nn
    def site(src, gamma, beta):
        bn = nn.BatchNorm2d(C).to(DEV)
        with torch.no_grad():
            bn.weight.copy_(gamma); bn.bias.copy_(beta); bn.running_mean.uniform_(-1, 1); bn.running_var.uniform_(0.5, 2)
        st = statbuf(2 * C)
        sv = nhwc(src, dt).double().reshape(-1, C)      # the statistics a producing conv would have accumulated, over 3 stripes
        st[:2 * C] = torch.cat([sv.sum(0), (sv * sv).sum(0)]) * 0.5
        st[2 * C:4 * C] = st[:2 * C] * 0.25
        st[8 * C:10 * C] = st[:2 * C] * 0.75            # stripe 4
        return bn, st
    for mom in (0.1, None):
        res = []
        for fused in (False, True):
            torch.manual_seed(5)
            bn_2, st2 = site(c2, g2, b2)
            bn_b, stb = site(sc_in, gb, bb)
            bn_2.momentum = bn_b.momentum = mom
            v2 = [torch.empty(C, device=DEV) for _ in range(4)]
            vb = [torch.empty(C, device=DEV) for _ in range(4)]
            o = torch.empty_like(outd)
            mk = torch.zeros_like(mask)
            if fused:
                f2 = ops.bn_fwd_fin(st2, bn_2, *v2)
                fb = ops.bn_fwd_fin(stb, bn_b, *vb) if bypass else None
                ops.block_tail_fwd_fin(c2d, f2, scd, fb, cnt, o, relu_mask=mk)
            else:
                for bn_, st_, v_ in ((bn_2, st2, v2),) + (((bn_b, stb, vb),) if bypass else ()):
                    ops.bn_finalize(st_, cnt, bn_.weight, bn_.bias, bn_.running_mean, bn_.running_var, bn_.num_batches_tracked,
                                    -1.0 if mom is None else mom, bn_.eps, *v_)
                ops.block_tail_fwd(c2d, v2[2], v2[0], v2[1], scd, vb[2] if bypass else None, vb[0] if bypass else None, vb[1] if bypass else None, o, relu_mask=mk)
            torch.cuda.synchronize()
            res.append((o, mk, v2, vb, bn_2, bn_b))
        (o0, m0, v20, vb0, bnA0, bnB0), (o1, m1, v21, vb1, bnA1, bnB1) = res
        assert torch.equal(o0, o1) and torch.equal(m0, m1)
        for a_, b_ in zip(v20, v21):
            assert torch.equal(a_, b_)
        assert torch.equal(bnA0.running_mean, bnA1.running_mean) and torch.equal(bnA0.running_var, bnA1.running_var) and int(bnA1.num_batches_tracked) == 1
        if bypass:
            for a_, b_ in zip(vb0, vb1):
                assert torch.equal(a_, b_)
            assert torch.equal(bnB0.running_mean, bnB1.running_mean) and torch.equal(bnB0.running_var, bnB1.running_var) and int(bnB1.num_batches_tracked) == 1
    # apply pass with both finalizes fused (no ubr_bn_bwd_finalize launch): bitwise the same gradients and dgamma / dbeta;
    # with and without the second gradient operand (separate kernel instantiations)
    for second in (go2d, None):
        r2f, rbf = statbuf(2 * C), statbuf(2 * C)
        ops.block_tail_bwd_reduce(go1d, second, junk, c2d, d(s2), d(t2), d(m2), d(i2), scd if bypass else None,
                                  d(mb) if bypass else None, d(ib) if bypass else None, r2f, rbf if bypass else None, relu_mask=mask)
        kf = torch.empty(4 * C, device=DEV)
        e = [torch.empty(C, device=DEV) for _ in range(4)]
        ops.bn_bwd_finalize(r2f, cnt, C, e[0], e[1], False, kf[:C], kf[C:2 * C])
        if bypass:
            ops.bn_bwd_finalize(rbf, cnt, C, e[2], e[3], False, kf[2 * C:3 * C], kf[3 * C:])
        ga_, gs_ = torch.empty_like(g_c2), torch.empty_like(g_sc)
        ops.block_tail_bwd_apply(go1d, second, junk, c2d, d(s2), d(t2), d(m2), d(i2), kf[:C], kf[C:2 * C],
                                 scd if bypass else None, d(sb) if bypass else None, d(mb) if bypass else None, d(ib) if bypass else None,
                                 kf[2 * C:3 * C] if bypass else None, kf[3 * C:] if bypass else None, ga_, gs_, relu_mask=mask)
        f = [torch.full((C,), float("nan"), device=DEV) for _ in range(4)]
        gb_, gt_ = torch.full_like(g_c2, float("nan")), torch.full_like(g_sc, float("nan"))
        ops.block_tail_bwd_apply_fin(go1d, second, mask, c2d, d(s2), d(t2), d(m2), d(i2), r2f, f[0], f[1],
                                     scd if bypass else None, d(sb) if bypass else None, d(mb) if bypass else None, d(ib) if bypass else None,
                                     rbf if bypass else None, f[2] if bypass else None, f[3] if bypass else None, cnt, gb_, gt_)
        torch.cuda.synchronize()
        assert torch.equal(gb_, ga_) and torch.equal(gt_, gs_)
        assert torch.equal(f[0], e[0]) and torch.equal(f[1], e[1])
        if bypass:
            assert torch.equal(f[2], e[2]) and torch.equal(f[3], e[3])
        else:
            # identity block: the skip gradient may be left to the consumer (g_sc = None)
            gn_ = torch.full_like(g_c2, float("nan"))
            ops.block_tail_bwd_apply_fin(go1d, second, mask, c2d, d(s2), d(t2), d(m2), d(i2), r2f, None, None,
                                         None, None, None, None, None, None, None, cnt, gn_, None)
            torch.cuda.synchronize()
            assert torch.equal(gn_, ga_)
        if second is not None:
            assert torch.equal(ga_, g_c2)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("shape", [(2, 32, 10, 12), (1, 16, 96, 128), (1, 512, 6, 10), (2, 192, 5, 8)])
def test_bn_backward(dt, relu, shape):
    N, C, H, W = shape
    c = rnd(dt, gen(N, C, H, W, seed=1))
    ga, ga2 = rnd(dt, gen(N, C, H, W, seed=2)), rnd(dt, gen(N, C, H, W, seed=3))
    gm, bt = _bn_vectors(C, 4)
    cr, gr, br = c.clone().requires_grad_(True), gm.clone().requires_grad_(True), bt.clone().requires_grad_(True)
    y = F.batch_norm(cr, None, None, gr, br, True, 0.1, 1e-5)
    a = F.relu(y) if relu else y
    a.backward(ga + ga2)
    m = c.mean(dim=(0, 2, 3))
    istd = 1 / torch.sqrt(c.var(dim=(0, 2, 3), unbiased=False) + 1e-5)
    sc, sh = gm * istd, bt
    d = lambda t: t.to(DEV)
    red = statbuf(2 * C)
    cd, gad, ga2d = nhwc(c, dt), nhwc(ga, dt), nhwc(ga2, dt)
    ops.bn_bwd_reduce(gad, ga2d, cd, d(sc), d(sh), d(m), d(istd), relu, red)
    k = torch.empty(2 * C, device=DEV)
    dg, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    ops.bn_bwd_finalize(red, N * H * W, C, dg, db, False, k[:C], k[C:])
    gc = torch.empty((N, H, W, C), dtype=dt, device=DEV)
    ops.bn_bwd_apply(gad, ga2d, cd, d(sc), d(sh), d(m), d(istd), relu, k[:C], k[C:], gc)
    torch.cuda.synchronize()
    t = 4 * tol(dt)
    close(nchw(gc), cr.grad, t, "g_c")
    close(dg.cpu(), gr.grad, t, "dgamma")
    close(db.cpu(), br.grad, t, "dbeta")
    # finalize fused into the apply pass: bitwise the same; single-gradient instantiation against its own two-launch form
    gcf, dgf, dbf = torch.full_like(gc, float("nan")), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    ops.bn_bwd_apply_fin(gad, ga2d, cd, d(sc), d(sh), d(m), d(istd), relu, red, N * H * W, dgf, dbf, gcf)
    torch.cuda.synchronize()
    assert torch.equal(gcf, gc) and torch.equal(dgf, dg) and torch.equal(dbf, db)
    red1 = statbuf(2 * C)
    ops.bn_bwd_reduce(gad, None, cd, d(sc), d(sh), d(m), d(istd), relu, red1)
    ops.bn_bwd_finalize(red1, N * H * W, C, dg, db, False, k[:C], k[C:])
    ops.bn_bwd_apply(gad, None, cd, d(sc), d(sh), d(m), d(istd), relu, k[:C], k[C:], gc)
    ops.bn_bwd_apply_fin(gad, None, cd, d(sc), d(sh), d(m), d(istd), relu, red1, N * H * W, dgf, dbf, gcf)
    torch.cuda.synchronize()
    assert torch.equal(gcf, gc) and torch.equal(dgf, dg) and torch.equal(dbf, db)
    cr.grad = None
    y = F.batch_norm(cr, None, None, gr, br, True, 0.1, 1e-5)
    (F.relu(y) if relu else y).backward(ga)
    close(nchw(gc), cr.grad, t, "g_c (one gradient operand)")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("stride", [2, 1])
def test_maxpool(dt, stride):
    N, C, H, W = 2, 16, 16, 24
    c = rnd(dt, gen(N, C, H, W, seed=1))
    sc, sh = gen(C, seed=2).abs() + 0.5, gen(C, seed=3) * 0.3
    # the kernels take the arg-max on the fp32 transformed values (forward and backward alike) and
    # round only what they store, so the reference pools the UNROUNDED transform
    xr = F.relu(c * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).requires_grad_(True)
    p = F.max_pool2d(xr, 3, stride, 1)
    gp = rnd(dt, gen(*p.shape, seed=4))
    ge = rnd(dt, gen(N, C, H, W, seed=5))
    p.backward(gp)
    aff = Affine(lo_zero(C), sc.to(DEV), sh.to(DEV), lo_zero(C))
    cd = nhwc(c, dt)
    pooled = torch.empty((N, p.shape[2], p.shape[3], C), dtype=dt, device=DEV)
    cat = torch.zeros((N, H, W, 2 * C), dtype=dt, device=DEV)
    xcopy = cat[..., C:] if stride == 2 else None
    amax = torch.empty(pooled.shape, dtype=torch.uint8, device=DEV) if stride == 2 else None
    ops.maxpool_fwd(cd, aff, pooled, xcopy, stride, argmax=amax)
    torch.cuda.synchronize()
    close(nchw(pooled), p.detach(), tol(dt), "pool fwd")
    if stride == 2:
        close(nchw(cat[..., C:]), xr.detach(), tol(dt), "pool xcopy")
        assert (cat[..., :C] == 0).all()
    gx = torch.empty((N, H, W, C), dtype=dt, device=DEV)
    ops.maxpool_bwd(cd, aff, nhwc(gp, dt), nhwc(ge, dt), gx, stride)
    torch.cuda.synchronize()
    # ties between equal maxima only occur at relu zeros, whose gradient the BN/ReLU mask kills
    # anyway; compare where the input is positive
    mask = (xr.detach() > 0).float()
    close(nchw(gx) * mask, (xr.grad + ge) * mask, tol(dt), "pool bwd")
    if stride == 2:
        # same gradient from the arg-max the forward saved (bit-identical to the re-scanning kernel: same first-max rule)
        gx2 = torch.empty_like(gx)
        ops.maxpool_bwd(cd, aff, nhwc(gp, dt), nhwc(ge, dt), gx2, stride, argmax=amax)
        torch.cuda.synchronize()
        assert torch.equal(gx2, gx)
        assert int(amax.max()) <= 8


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("C", [3, 4])
def test_logsoftmax_backward(dt, C):
    N, H, W = 2, 16, 24
    z = gen(N, C, H, W, seed=1).requires_grad_(True)
    lp = F.log_softmax(z, 1)
    g = gen(N, C, H, W, seed=2)
    lp.backward(g)
    out = torch.full((N, H, W, 16), 9.0, dtype=dt, device=DEV)
    ops.logsoftmax_bwd(g.to(DEV), lp.detach().to(DEV), out)
    torch.cuda.synchronize()
    close(nchw(out[..., :C]), z.grad, tol(dt), "logsoftmax bwd")
    assert (out[..., C:].float() == 0).all()


def test_pixelwise_nll_and_confusion():
    from ubresnet_amd.training.pixelwise_nllloss import PixelWiseNLLLoss
    from ubresnet_amd import metrics
    from oracle import uresnet_oracle as O
    N, C, H, W = 2, 3, 32, 32
    lp = F.log_softmax(gen(N, C, H, W, seed=1), 1)
    tg = torch.randint(0, C, (N, H, W), generator=torch.Generator().manual_seed(2))
    tg[0, :2, :] = -100
    pw = gen(N, H, W, seed=3).abs() + 0.5
    lr = lp.clone().requires_grad_(True)
    ref = O.pixelwise_nll(lr, tg, pw)
    ref.backward()
    crit = PixelWiseNLLLoss()
    ld = lp.to(DEV).requires_grad_(True)
    loss = crit.forward(ld, tg.to(DEV), pw.to(DEV))
    loss.backward()
    assert abs(loss.item() - ref.item()) <= 1e-6 * abs(ref.item())
    close(ld.grad.cpu(), lr.grad, 1e-6, "nll grad")
    # class weights
    cw = torch.tensor([0.5, 2.0, 1.5])
    ref2 = O.pixelwise_nll(lp, tg, pw, cw)
    loss2 = PixelWiseNLLLoss(weight=cw)(lp.to(DEV), tg.to(DEV), pw.to(DEV))
    assert abs(loss2.item() - ref2.item()) <= 1e-6 * abs(ref2.item())
    # accuracy / confusion: bit-exact integers
    tg2 = tg.clamp(min=0)
    cm = metrics.confusion_matrix(lp.to(DEV), tg2.to(DEV)).cpu()
    assert torch.equal(cm, O.confusion_matrix(lp, tg2))
    acc, racc = metrics.accuracy(lp.to(DEV), tg2.to(DEV)), O.accuracy(lp, tg2)
    assert all(abs(a - b) < 1e-9 for a, b in zip(acc, racc))


def test_out_of_range_labels_raise():
    """F.nll_loss device-asserts on a target outside [0,C) other than ignore_index (training/pixelwise_nllloss.py:51); the
    HIP loss counts them and raises RuntimeError -- at the next loss call by default (asynchronous, no sync in the step)"""
    from ubresnet_amd.training import pixelwise_nllloss as PL
    lp = torch.log_softmax(torch.randn(2, 3, 8, 8, device="cuda", requires_grad=True), 1)      # a loss that will be back-propagated
    tg = torch.randint(0, 3, (2, 8, 8), device="cuda")
    pw = torch.ones(2, 8, 8, device="cuda")
    crit = PL.PixelWiseNLLLoss()
    good = crit(lp, tg, pw)
    tg_ign = tg.clone(); tg_ign[0, 0, :4] = -100          # ignore_index is not an error
    crit(lp, tg_ign, pw)
    torch.cuda.synchronize()
    crit(lp, tg, pw)                                       # polls the two finished checks: clean
    bad = tg.clone(); bad[1, 2, 3] = 3                     # the 4-class cosmic labels fed to a 3-class network
    loss = crit(lp, bad, pw)                               # value is defined (pixel contributes 0), the report is pending
    assert torch.isfinite(loss)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="outside"):
        crit(lp, tg, pw)
    old = PL._label_check.mode
    PL._label_check.mode = "sync"
    try:
        with pytest.raises(RuntimeError, match="outside"):
            crit(lp, bad, pw)
    finally:
        PL._label_check.mode = old
        PL._label_check.pending.clear()
    assert abs(float(crit(lp, tg, pw)) - float(good)) < 1e-7
    # a bad label in the LAST batch has no "next call": flush() (end of epoch / before a checkpoint) reports it
    crit(lp, bad, pw).backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="outside"):
        crit.flush()
    crit.flush()                                           # nothing pending: silent
    # a loss nobody back-propagates (validation) reports immediately, as F.nll_loss would
    with torch.no_grad(), pytest.raises(RuntimeError, match="outside"):
        crit(lp.detach(), bad, pw)
    PL._label_check.pending.clear()


@pytest.mark.parametrize("dt", DTS + [torch.float16])
@pytest.mark.parametrize("act", [1, 2, 3])
@pytest.mark.parametrize("ksz", [3, 7])
def test_conv_inference_epilogue(dt, act, ksz):
    """ubr_conv_desc.act: out = relu?( relu?(conv + bias) + addend ) -- the BasicBlock tail of the inference schedule
    (models/common_layers.py:47-56 with eval-mode BatchNorm folded into weights and bias); 7x7 = conv10 of the head
    (models/ub_uresnet.py:60), which runs the row-stationary thin kernel in fp16 / bf16"""
    N, H, W, Cin, Cout = (2, 16, 32, 32, 32) if ksz == 3 else (2, 32, 32, 16, 16)
    k2 = ksz * ksz
    x = rnd(dt, gen(N, Cin, H, W, seed=1))
    w = gen(Cout, Cin, ksz, ksz, seed=2, scale=(2.0 / (k2 * Cin)) ** 0.5)
    b = gen(Cout, seed=3) * 0.5
    ad = rnd(dt, gen(N, Cout, H, W, seed=6))
    ref = F.conv2d(x, rnd(dt, w), b, 1, ksz // 2)
    if act & 1:
        ref = F.relu(ref)
    ref = ref + ad
    if act & 2:
        ref = F.relu(ref)
    y = torch.empty((N, H, W, Cout), dtype=dt, device=DEV)
    wp = ops.pack_weights(w.to(DEV), dt, Cout, Cin, Cin * k2, k2, k2)
    ops.conv(nhwc(x, dt), wp, y, ops.conv_taps(ksz, 1, ksz // 2), Cout, bias=b.to(DEV), addend=nhwc(ad, dt), act=act)
    torch.cuda.synchronize()
    close(nchw(y), ref, tol(dt), "conv act=%d" % act)


@pytest.mark.parametrize("dt", DTS)
def test_batched_pack_and_bn_fold(dt):
    """ubr_pack_weights_batched (coalesced LDS transpose; both dense orientations, the gather path, an output-row
    scale) against ubr_pack_weights item by item, and ubr_bn_fold_batched against its fp64 formula"""
    import struct
    from ubresnet_amd import _lib as L
    cpu = L.chans_per_unit(dt)
    cases = []      # (weight, M, Kvalid, Kpad, sm, sk, ntaps, tap_stride, src_offset, scaled)
    for (d0, d1, k, seed) in ((48, 24, 3, 1), (16, 40, 7, 2), (3, 16, 7, 3), (64, 64, 1, 4), (32, 16, 4, 5)):
        w = gen(d0, d1, k, k, seed=seed).to(DEV)
        kk = k * k
        cases.append((w, d0, d1, None, d1 * kk, kk, kk, 1, 0, True))         # Conv2d forward orientation (mode A), scaled rows
        cases.append((w, d1, d0, None, kk, d1 * kk, kk, 1, 0, False))        # data-gradient orientation (mode B)
    w7 = gen(16, 2, 7, 7, seed=9).to(DEV)
    cases.append((w7, 16, 7, 16, 2 * 49, 1, 7, 7, 49, True))                 # column-expanded stem, plane 1 (gather path)
    items, dsts, refs, keep = b"", [], [], []
    for (w, M, Kv, Kpad, sm, sk, ntaps, tstride, soff, scaled) in cases:
        Mpad = (M + 15) // 16 * 16
        Kp = Kpad if Kpad is not None else (Kv + cpu - 1) // cpu * cpu
        dst = torch.full((ntaps, Kp // cpu, Mpad, cpu), 5.0, dtype=dt, device=DEV)
        sc = (gen(M, seed=11).abs() + 0.5).to(DEV) if scaled else None
        keep.append(sc)            # the table holds raw device addresses: the scale vectors must outlive the launch
        items += struct.pack("<QQqqqQiiiiii", w.data_ptr() + 4 * soff, dst.data_ptr(), sm, sk, tstride, sc.data_ptr() if scaled else 0,
                             M, Mpad, Kv, Kp // cpu, ntaps, 0)
        dsts.append(dst)
        wsrc = w
        if scaled:     # scaling rows of the source == scaling output rows (M indexes dim 0 in the scaled cases)
            wsrc = (w * sc.view(-1, 1, 1, 1)).contiguous()
        refs.append(ops.pack_weights(wsrc, dt, M, Kv, sm, sk, ntaps, tapidx=[t * tstride for t in range(ntaps)], Kpad=Kp, src_offset=soff))
    tbl = torch.frombuffer(bytearray(items), dtype=torch.uint8).to(DEV)
    L.check(L.lib().ubr_pack_weights_batched(L.dtype_id(dt), tbl.data_ptr(), len(cases), L.stream_ptr()), "pack")
    torch.cuda.synchronize()
    for i, (got, ref) in enumerate(zip(dsts, refs)):
        if dt == torch.float32 and cases[i][-1]:
            assert torch.allclose(got, ref, rtol=2e-7, atol=0), "item %d" % i      # w*s rounded once on either side
        elif cases[i][-1]:
            assert (got.float() - ref.float()).abs().max() <= 2 ** -7 * ref.float().abs().max(), "item %d" % i   # 1 bf16 ulp
        else:
            assert torch.equal(got, ref), "item %d" % i
    # BatchNorm fold
    Cn = 40
    gam, bet, rm = gen(Cn, seed=1).to(DEV), gen(Cn, seed=2).to(DEV), gen(Cn, seed=3).to(DEV)
    rv, cb = (gen(Cn, seed=4).abs() + 0.1).to(DEV), gen(Cn, seed=5).to(DEV)
    out = torch.zeros(4 * Cn, device=DEV)
    it = struct.pack("<QQQQQQQif", gam.data_ptr(), bet.data_ptr(), rm.data_ptr(), rv.data_ptr(), cb.data_ptr(), out.data_ptr(), out[Cn:].data_ptr(), Cn, 1e-5)
    it += struct.pack("<QQQQQQQif", gam.data_ptr(), bet.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0, out[2 * Cn:].data_ptr(), out[3 * Cn:].data_ptr(), Cn, 1e-5)
    t2 = torch.frombuffer(bytearray(it), dtype=torch.uint8).to(DEV)
    L.check(L.lib().ubr_bn_fold_batched(t2.data_ptr(), 2, L.stream_ptr()), "fold")
    torch.cuda.synchronize()
    s = gam.double() / torch.sqrt(rv.double() + 1e-5)
    assert torch.allclose(out[:Cn].double(), s, rtol=1e-6) and torch.allclose(out[2 * Cn:3 * Cn].double(), s, rtol=1e-6)
    assert torch.allclose(out[Cn:2 * Cn].double(), (cb.double() - rm.double()) * s + bet.double(), rtol=1e-6, atol=1e-6)
    assert torch.allclose(out[3 * Cn:].double(), (0 - rm.double()) * s + bet.double(), rtol=1e-6, atol=1e-6)
