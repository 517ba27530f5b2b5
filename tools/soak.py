import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubresnet_amd.models.ub_uresnet import UResNet
from ubresnet_amd.training.pixelwise_nllloss import PixelWiseNLLLoss
from ubresnet_amd.optim import FlatAdam
from ubresnet_amd import synthetic, metrics
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = UResNet(num_classes=3, input_channels=1, inplanes=16).to(dev); model.compute_dtype = torch.bfloat16; model.train()
crit = PixelWiseNLLLoss(); opt = FlatAdam(model, lr=1e-3, weight_decay=1e-4)
ld = synthetic.SyntheticLArCVDataset(height=512, width=512, tag="train", nentries=64); ld.start(16)
st = synthetic.DeviceStager(ld, 16, 512, 512, tag="train")
t0 = time.perf_counter(); losses = []
for i in range(300):
    x, lab, wgt = st.next()
    out = model.forward(x); loss = crit.forward(out, lab, wgt); opt.zero_grad(); loss.backward(); opt.step()
    if i % 50 == 0 or i == 299:
        losses.append(round(loss.item(), 4))
        print(i, losses[-1], "mem GB %.2f" % (torch.cuda.max_memory_allocated() / 2**30), "acc", [round(a, 1) for a in metrics.accuracy(out.detach(), lab)], flush=True)
torch.cuda.synchronize(); print("300 steps in %.1f s; finite params: %s" % (time.perf_counter() - t0, all(torch.isfinite(p).all().item() for p in model.parameters())))
