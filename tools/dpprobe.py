"""1-rank RCCL rehearsal of the data-parallel step with host-side timing of each segment (run under torchrun,
UBR_FORCE_REDUCER=1).  UBR_BUCKET_MB sets the bucket size."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from ubresnet_amd.models.ub_uresnet import UResNet
from ubresnet_amd.training.pixelwise_nllloss import PixelWiseNLLLoss
from ubresnet_amd import synthetic
from ubresnet_amd.dist import GradAllReducer
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
torch.manual_seed(0)
model = UResNet(num_classes=3, input_channels=1, inplanes=16).to(dev); model.compute_dtype = torch.bfloat16; model.train()
crit = PixelWiseNLLLoss()
opt = torch.optim.Adam(model.parameters(), lr=1e-5, weight_decay=1e-4, fused=True)
if os.environ.get("UBR_FAKE_ALLREDUCE") == "1":
    class _W:
        def wait(self): pass
    dist.all_reduce = lambda t, op=None, group=None, async_op=False: _W()     # keep every stream dependency, drop the collective
red = GradAllReducer(model, bucket_bytes=int(float(os.environ.get("UBR_BUCKET_MB", "16")) * (1 << 20)))
x, lab, wgt = synthetic.make_batch(16, 512, 512, seed0=1000)
x, lab, wgt = torch.from_numpy(x).to(dev), torch.from_numpy(lab).to(dev), torch.from_numpy(wgt).to(dev)
def step(tm):
    t0 = time.perf_counter(); out = model.forward(x); loss = crit.forward(out, lab, wgt)
    t1 = time.perf_counter(); opt.zero_grad(); loss.backward()
    t2 = time.perf_counter(); red.finish()
    t3 = time.perf_counter(); opt.step(); t4 = time.perf_counter()
    for i, v in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)): tm[i] += v
for _ in range(3): step([0] * 4)
torch.cuda.synchronize()
tm = [0] * 4; t0 = time.perf_counter()
for _ in range(10): step(tm)
ti = time.perf_counter() - t0
torch.cuda.synchronize(); tt = time.perf_counter() - t0
print("bucket %s MB: issue %.2f ms/step, total %.2f ms/step; host fwd %.2f bwd %.2f finish %.2f opt %.2f" % (
    os.environ.get("UBR_BUCKET_MB", "16"), ti * 100, tt * 100, *(v * 100 for v in tm)))
dist.destroy_process_group()
