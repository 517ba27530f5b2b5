"""whole-train-step hipGraph capture experiment (one GPU, no collective)"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubresnet_amd.models.ub_uresnet import UResNet
from ubresnet_amd.training.pixelwise_nllloss import PixelWiseNLLLoss
from ubresnet_amd.optim import FlatAdam
from ubresnet_amd import synthetic
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = UResNet(num_classes=3, input_channels=1, inplanes=16).to(dev); model.compute_dtype = torch.bfloat16; model.train()
crit = PixelWiseNLLLoss()
opt = FlatAdam(model, lr=1e-5, weight_decay=1e-4)
x, lab, wgt = synthetic.make_batch(16, 512, 512, seed0=1000)
x, lab, wgt = torch.from_numpy(x).to(dev), torch.from_numpy(lab).to(dev), torch.from_numpy(wgt).to(dev)
def step():
    out = model.forward(x); loss = crit.forward(out, lab, wgt); opt.zero_grad(); loss.backward(); opt.step(); return loss
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): l = step()
torch.cuda.synchronize(); print("eager %.2f ms/step loss %.5f" % ((time.perf_counter() - t0) * 100, l.item()))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    sl = step()
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): g.replay()
torch.cuda.synchronize(); print("graph %.2f ms/step loss %.5f" % ((time.perf_counter() - t0) * 100, sl.item()))
