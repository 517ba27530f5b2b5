import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle.uresnet_oracle as O
from ubresnet_amd import synthetic
from ubresnet_amd.models.ub_uresnet import UResNet
sd = O.seeded_state_dict(O.uresnet_schema(3, 1, 16, 16), 42)
x, lab, wgt = synthetic.make_batch(16, 512, 512, 1000)
xt = torch.from_numpy(x).cuda()
m = UResNet(num_classes=3, input_channels=1, inplanes=16); m.load_state_dict(sd); m = m.cuda(); m.train() if 'train' in sys.argv else m.eval()
with torch.no_grad():
    ref = m(xt); m.compute_dtype = torch.bfloat16; out = m(xt); m.compute_dtype = torch.float16; o16 = m(xt)
d = (out - ref).abs()
print("bf16: max abs diff %.4f, mean %.5f, p99.9 %.4f" % (d.max().item(), d.mean().item(), torch.quantile(d.reshape(-1)[::97].float(), 0.999).item()))
d16 = (o16 - ref).abs(); print("fp16: max abs diff %.5f mean %.6f" % (d16.max().item(), d16.mean().item()))
top2 = torch.topk(ref, 2, dim=1)[0]; mg = (top2[:, 0] - top2[:, 1]).reshape(-1)
a, b, c = ref.argmax(1).reshape(-1), out.argmax(1).reshape(-1), o16.argmax(1).reshape(-1)
print("class counts fp32", torch.bincount(a, minlength=3).tolist(), "bf16", torch.bincount(b, minlength=3).tolist())
for th in (0.0, 0.05, 0.1, 0.2, 0.5):
    s = mg > th
    cm = torch.bincount(a[s] * 3 + b[s], minlength=9).reshape(3, 3).cpu()
    cm16 = torch.bincount(a[s] * 3 + c[s], minlength=9).reshape(3, 3).cpu()
    print("margin>%.2f: %.2f%% px, bf16 IoU %s, fp16 IoU %s" % (th, 100 * s.float().mean().item(), [round(float(v), 4) for v in O.iou_from_confusion(cm)], [round(float(v), 4) for v in O.iou_from_confusion(cm16)]))
print("logp range", ref.min().item(), ref.max().item(), "margin quantiles", [round(torch.quantile(mg[::97].float(), q).item(), 3) for q in (0.01, 0.1, 0.5, 0.9)])
