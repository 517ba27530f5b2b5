"""conv11 data gradient exactly as the executor issues it: K = num_classes zero-padded to 16."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
if torch.cuda.is_available():
    from ubresnet_amd import ops, _lib as L
    import ctypes as C


@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 96, 128), (2, 64, 96)])
@pytest.mark.parametrize("ncls", [3, 4])
def test_conv11_dgrad_padded(shape, ncls):
    N, H, W = shape
    g = torch.Generator().manual_seed(1)
    w = torch.randn(ncls, 16, 7, 7, generator=g) * 0.1
    gl = torch.randn(N, ncls, H, W, generator=g)
    a = torch.randn(N, 16, H, W, generator=g).requires_grad_(True)
    F.conv2d(a, w, None, 1, 3).backward(gl)
    gld = torch.zeros((N, H, W, 16), device="cuda")
    gld[..., :ncls] = gl.permute(0, 2, 3, 1).cuda()
    wd = w.cuda()
    dst = torch.empty((49, 4, 16, 4), device="cuda")
    idx = (C.c_int32 * 49)(*range(49))
    L.check(L.lib().ubr_pack_weights(0, wd.data_ptr(), dst.data_ptr(), 16, 16, ncls, 16, 49, 16 * 49, 49, idx, L.stream_ptr()))
    out = torch.empty((N, H, W, 16), device="cuda")
    ops.conv(gld, dst, out, ops.conv_dgrad_taps_s1(7, 1, 3), 16)
    torch.cuda.synchronize()
    got = out.permute(0, 3, 1, 2).cpu()
    err = (got - a.grad).abs()
    rel = err.max().item() / a.grad.abs().max().item()
    if rel > 2e-5:
        bad = (err > 1e-4 * a.grad.abs().max()).nonzero()
        print("bad count", bad.shape[0], "first", bad[:10].tolist(), "last", bad[-10:].tolist())
    assert rel <= 2e-5, "conv11 dgrad rel err %.3e" % rel
