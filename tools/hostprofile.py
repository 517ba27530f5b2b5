"""cProfile of the host side of a train step (tiny tensors, so the GPU never limits): where the Python/ctypes time goes."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ubresnet_amd.models.ub_uresnet import UResNet
from ubresnet_amd.training.pixelwise_nllloss import PixelWiseNLLLoss
from ubresnet_amd import synthetic
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = UResNet(num_classes=3, input_channels=1, inplanes=16).to(dev); model.compute_dtype = torch.bfloat16; model.train()
crit = PixelWiseNLLLoss()
opt = torch.optim.Adam(model.parameters(), lr=1e-5, weight_decay=1e-4, fused=True)
x, lab, wgt = synthetic.make_batch(1, 64, 64, seed0=1000)
x, lab, wgt = torch.from_numpy(x).to(dev), torch.from_numpy(lab).to(dev), torch.from_numpy(wgt).to(dev)
def step():
    out = model.forward(x); loss = crit.forward(out, lab, wgt); opt.zero_grad(); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print("host-bound step: %.2f ms" % ((time.perf_counter() - t0) * 50))
torch.autograd.set_multithreading_enabled(False)      # run the custom backward in this thread so cProfile sees it
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(32)
