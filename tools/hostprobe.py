"""host-side issue time of the train step (what the launch plans of ubresnet_amd/plan.py remove): UBR_PLAN=0/1"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubresnet_amd.models.ub_uresnet import UResNet
from ubresnet_amd.training.pixelwise_nllloss import PixelWiseNLLLoss
from ubresnet_amd import synthetic
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = UResNet(num_classes=3, input_channels=1, inplanes=16).to(dev); model.compute_dtype = torch.bfloat16; model.train()
crit = PixelWiseNLLLoss()
from ubresnet_amd.optim import FlatAdam
opt = FlatAdam(model, lr=1e-5, weight_decay=1e-4)
x, lab, wgt = synthetic.make_batch(16, 512, 512, seed0=1000)
x, lab, wgt = torch.from_numpy(x).to(dev), torch.from_numpy(lab).to(dev), torch.from_numpy(wgt).to(dev)
def step(tm):
    t0 = time.perf_counter(); out = model.forward(x); loss = crit.forward(out, lab, wgt)
    t1 = time.perf_counter(); opt.zero_grad(); loss.backward()
    t2 = time.perf_counter(); opt.step(); t3 = time.perf_counter()
    tm[0] += t1 - t0; tm[1] += t2 - t1; tm[2] += t3 - t2
for _ in range(3): step([0, 0, 0])
torch.cuda.synchronize()
tm = [0, 0, 0]; t0 = time.perf_counter()
for _ in range(10): step(tm)
ti = time.perf_counter() - t0
torch.cuda.synchronize(); tt = time.perf_counter() - t0
# issue time from an idle GPU (no launch-queue back-pressure: once the host is faster than the GPU the loop above measures the GPU)
ti2 = 0.0; tmi = [0, 0, 0]
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(tmi); ti2 += time.perf_counter() - t0
torch.cuda.synchronize()
print("from idle: issue %.2f ms/step; host fwd %.2f bwd %.2f opt %.2f" % (ti2 * 100, tmi[0] * 100, tmi[1] * 100, tmi[2] * 100))
print("UBR_PLAN=%s" % os.environ.get("UBR_PLAN", "1"), "issue %.2f ms/step, total %.2f ms/step; host fwd %.2f bwd %.2f opt %.2f" % (ti * 100, tt * 100, tm[0] * 100, tm[1] * 100, tm[2] * 100))
