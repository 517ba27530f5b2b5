import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from ubresnet_amd import ops
from ubresnet_amd.ops import Affine
torch.manual_seed(0)
def check(N, H, W, Cin, Cout, k, dt, xf=False):
    x = torch.randn(N, Cin, H, W); g = torch.randn(N, Cout, H, W)
    xq, gq = x.to(dt).float(), g.to(dt).float()
    xr = xq
    if xf:
        sc, sh = torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.3
        xr = F.relu(xq * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    ref = torch.nn.grad.conv2d_weight(xr.double(), (Cout, Cin, k, k), gq.double(), padding=k // 2).float()
    xd = xq.permute(0, 2, 3, 1).contiguous().to(dt).cuda(); gd = gq.permute(0, 2, 3, 1).contiguous().to(dt).cuda()
    dW = torch.empty(Cout, Cin, k, k, device="cuda")
    ws = ops.WgradWorkspace()
    aff = Affine(torch.zeros(Cin).cuda(), sc.cuda(), sh.cuda(), torch.zeros(Cin).cuda()) if xf else None
    ops.wgrad(xd, gd, ops.conv_taps(k, 1, k // 2), dW, Cin * k * k, k * k, Cout, Cin, ws, xf=aff)
    torch.cuda.synchronize()
    err = (dW.cpu() - ref).abs().max().item() / ref.abs().max().item()
    print("N%d %dx%d %d->%d k%d %s xf=%d: rel err %.3e %s" % (N, H, W, Cin, Cout, k, str(dt)[6:], xf, err, "OK" if err < (2e-5 if dt == torch.float32 else 2e-2) else "FAIL"))
for dt in (torch.float32, torch.bfloat16):
    check(2, 128, 128, 64, 64, 3, dt)
    check(2, 128, 128, 64, 64, 3, dt, xf=True)
    check(2, 256, 256, 32, 32, 3, dt)
    check(2, 512, 512, 16, 16, 3, dt, xf=True)
    check(2, 64, 64, 128, 128, 3, dt)
    check(1, 256, 256, 32, 16, 1, dt)
    check(2, 128, 128, 16, 16, 7, dt)
