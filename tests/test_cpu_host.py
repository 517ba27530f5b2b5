"""CPU-side tests: the C-ABI library loads and exports every symbol include/ubresnet_hip.h declares,
the mirror modules keep the reference's constructor/state_dict surface, the tap geometry is right,
the synthetic loader speaks the larcvdataset contract, and the product fails loudly without a GPU."""
import os
import sys
import re

import numpy as np
import pytest
import torch

from oracle import uresnet_oracle as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from ubresnet_amd import _lib
    hdr = open(os.path.join(REPO, "include", "ubresnet_hip.h")).read()
    declared = set(re.findall(r"\b(ubr_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"ubr_wgrad_workspace"}          # mentioned in a comment only
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.lib()
    for s in sorted(declared):
        assert hasattr(lib, s), "libubresnet_hip.so does not export %s" % s
    assert lib.ubr_version() >= 1


def test_struct_layouts_match_header():
    """ctypes mirrors of the descriptor structs have the C sizes (hipcc compiled the same header)."""
    import ctypes as C
    from ubresnet_amd import _lib as L
    assert C.sizeof(L.Tensor) == 32 and C.sizeof(L.ChanAffine) == 32
    # int32 x5, Tensor, ChanAffine, ptr, int32 x3, 3x64 bytes, int32 x5, Tensor x2, ptr x2, int32 x2
    assert C.sizeof(L.ConvDesc) == 20 + 4 + 32 + 32 + 8 + 12 + 192 + 20 + 64 + 16 + 8 + 8 + 8 + 32 + 32 + 40 + 64 + 8
    assert C.sizeof(L.WgradDesc) % 8 == 0


def test_uresnet_surface_matches_reference(golden_dir):
    from ubresnet_amd.models.ub_uresnet import UResNet
    m = UResNet(num_classes=3, input_channels=1, inplanes=16)
    lines = open(os.path.join(golden_dir, "state_dict_keys_uresnet_ip16.txt")).read().split("\n")[:-1]
    want = [(l.split(" ")[0], l.split(" ")[1]) for l in lines]
    got = [(k, "x".join(map(str, v.shape)) or "scalar") for k, v in m.state_dict().items()]
    assert got == want
    # positional constructor order of the reference (models/ub_uresnet.py:31)
    m2 = UResNet(4, 1, 16, 16, False)
    assert m2.conv11.out_channels == 4 and m2.conv1.in_channels == 1
    # init scale (models/ub_uresnet.py:73-79): std = sqrt(2/(k*k*out_channels)); BN gamma=1 beta=0
    w = m.enc_layer3.res1.conv1.weight
    assert abs(w.std().item() - (2.0 / (9 * w.shape[0])) ** 0.5) < 0.1 * (2.0 / (9 * w.shape[0])) ** 0.5
    assert float(m.bn10.weight.min()) == 1.0 and float(m.bn10.bias.abs().max()) == 0.0
    # checkpoints written with a DataParallel "module." prefix load after stripping (deploy/ubresnet_funcs.py:52-66)
    sd = {"module." + k: v for k, v in m.state_dict().items()}
    m2 = UResNet(3, 1, 16)
    m2.load_state_dict({k[len("module."):]: v for k, v in sd.items()})
    # nn.Module protocol
    assert sum(p.numel() for p in m.parameters()) == 18100931
    m.eval(); m.train()
    assert "UResNet" in repr(m) and "enc_layer5" in repr(m)


def test_no_cpu_fallback():
    from ubresnet_amd.models.ub_uresnet import UResNet
    from ubresnet_amd.training.pixelwise_nllloss import PixelWiseNLLLoss
    m = UResNet(num_classes=3, input_channels=1, inplanes=16)
    with pytest.raises(RuntimeError, match="ROCm device"):
        m(torch.zeros(1, 1, 32, 32))
    with pytest.raises(RuntimeError, match="parameter holder"):
        m.conv1(torch.zeros(1, 1, 32, 32))
    with pytest.raises(RuntimeError, match="fused HIP graph"):
        m.enc_layer1(torch.zeros(1, 16, 32, 32))
    with pytest.raises(RuntimeError, match="ROCm device"):
        PixelWiseNLLLoss()(torch.zeros(1, 3, 4, 4), torch.zeros(1, 4, 4, dtype=torch.int64), torch.ones(1, 4, 4))
    with pytest.raises(ValueError):
        UResNet(num_classes=3, input_channels=1, inplanes=24)


def test_import_styles():
    """both import styles of the reference: sys.path += UBRESNET_MODELDIR and package import"""
    import subprocess, sys
    code = ("import sys; sys.path.append(%r); from ub_uresnet import UResNet; import common_layers; "
            "from ubresnet_amd.models.ub_uresnet import UResNet as U2; assert UResNet is U2; print('ok')"
            % os.path.join(REPO, "ubresnet_amd", "models"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp")
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr


def test_tap_geometry_against_torch():
    """phase decomposition of transposed convs: taps reproduce F.conv_transpose2d / conv dgrad on CPU"""
    import torch.nn.functional as F
    from ubresnet_amd import ops
    g = torch.Generator().manual_seed(0)
    for k, pad, s, H in ((4, 1, 2, 6), (3, 1, 2, 8), (1, 0, 2, 8)):
        x = torch.randn(1, 1, H, H, generator=g)
        w = torch.randn(1, 1, k, k, generator=g)
        op = 1 if k != 4 else 0
        ref = F.conv_transpose2d(x, w, None, s, pad, output_padding=op)
        out = torch.zeros_like(ref)
        xp = F.pad(x, (4, 4, 4, 4))
        for ry in range(s):
            for rx in range(s):
                for dy, dx, t in ops.transposed_phase_taps(k, 1, pad, s, ry, rx):
                    ph = out[:, :, ry::s, rx::s]
                    n_y, n_x = ph.shape[2], ph.shape[3]
                    ph += xp[:, :, 4 + dy:4 + dy + n_y, 4 + dx:4 + dx + n_x] * w.reshape(-1)[t]
        assert (out - ref).abs().max() < 1e-5, (k, pad, s)
    # stride-1 dgrad taps
    x = torch.randn(1, 1, 9, 9, generator=g, requires_grad=True)
    w = torch.randn(1, 1, 3, 3, generator=g)
    gy = torch.randn(1, 1, 9, 9, generator=g)
    F.conv2d(x, w, None, 1, 1).backward(gy)
    gp = F.pad(gy, (2, 2, 2, 2))
    acc = torch.zeros(1, 1, 9, 9)
    for dy, dx, t in ops.conv_dgrad_taps_s1(3, 1, 1):
        acc += gp[:, :, 2 + dy:11 + dy, 2 + dx:11 + dx] * w.reshape(-1)[t]
    assert (acc - x.grad).abs().max() < 1e-5


def test_synthetic_loader_contract():
    from ubresnet_amd.synthetic import SyntheticLArCVDataset, make_batch
    ld = SyntheticLArCVDataset(height=64, width=96, tag="train", nentries=10)
    with pytest.raises(RuntimeError):
        ld[0]
    ld.start(3)
    d = ld[0]
    assert set(d) == {"source_train", "label_train", "weight_train"}
    assert all(v.dtype == np.float32 and v.shape == (3 * 64 * 96,) for v in d.values())
    # prep_data's reshapes (training/train_ubresnet2018_wlarcv2.py:600-603)
    src = torch.from_numpy(d["source_train"].reshape((3, 1, 64, 96)))
    lab = torch.from_numpy(d["label_train"].reshape((3, 64, 96)).astype(np.int64))
    assert src.shape == (3, 1, 64, 96) and int(lab.max()) <= 2 and int(lab.min()) == 0
    a, l, w = make_batch(3, 64, 96, 1000)
    assert np.array_equal(a.reshape(-1), d["source_train"])           # deterministic per entry seed
    occ = float((a != 0).mean())
    assert 0.002 < occ < 0.2
    assert len(ld) == 10
    ld.stop()
    a512, _, _ = make_batch(1, 512, 512, 1000)
    assert 0.005 < float((a512 != 0).mean()) < 0.05                    # ~1-3 % occupancy at the benchmark size


def test_grad_completion_order_covers_all_parameters():
    from ubresnet_amd.models.ub_uresnet import UResNet
    m = UResNet(num_classes=3, input_channels=1, inplanes=16)
    names = [n for n, _ in m._grad_completion_order()]
    assert sorted(names) == sorted(n for n, _ in m.named_parameters())
    assert names[0] == "conv11.weight" and names[-1] == "conv1.bias"


def test_shard_range():
    from ubresnet_amd.dist import shard_range
    assert [shard_range(128, r, 8) for r in (0, 7)] == [(0, 16), (112, 128)]
    with pytest.raises(ValueError):
        shard_range(10, 0, 4)


def test_header_is_plain_c(tmp_path):
    """the drop-in boundary is a C ABI: include/ubresnet_hip.h must compile as C99 with nothing but <stdint.h>"""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "ubresnet_hip.h")
    src = tmp_path / "hdr.c"
    src.write_text('#include "%s"\nint main(void) { ubr_conv_desc d; (void)d; return (int)sizeof(ubr_wgrad_desc) * 0; }\n' % hdr)
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-fsyntax-only", str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_aspp_resnet1_alias_both_import_styles():
    """the reference's ASPP train scripts import `ASPP_ResNet1` (training/Sem_Seg_ASPP_ResNet1.py:43,
    training/grid_scripts/train_aspp_wlarcv1_tuftsgrid.py:38): both styles resolve to the one module"""
    import subprocess, sys
    code = ("import sys; sys.path.append(%r); sys.path.append(%r); from ASPP_ResNet1 import ASPP_ResNet; "
            "from models.ASPP_ResNet1 import ASPP_ResNet as B; from ubresnet_amd.models.ASPP_ResNet1 import ASPP_ResNet as C; "
            "from ubresnet_amd.models.ASPP_ResNet import ASPP_ResNet as D; assert ASPP_ResNet is B is C is D; "
            "m = ASPP_ResNet(3, 3, 16, False); print('ok', len(m.state_dict()))"
            % (os.path.join(REPO, "ubresnet_amd", "models"), os.path.join(REPO, "ubresnet_amd")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp")
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr


def test_data_parallel_wrappers_get_gradients_through_autograd(monkeypatch):
    """The reference wraps the model in nn.DataParallel (training/train_ubresnet2018_wlarcv2.py:99,103).  Its replicas hold
    non-leaf broadcast copies of the parameters and DDP relies on AccumulateGrad hooks: under either wrapper the fused autograd
    node must hand its gradients BACK THROUGH autograd (compatibility path) instead of installing .grad views that the
    wrappers never see.  The executor is replaced by a stand-in here (no GPU): what is tested is the gradient routing."""
    from ubresnet_amd import engine as E
    from ubresnet_amd.models.ub_uresnet import UResNet
    m = UResNet(num_classes=3, input_channels=1, inplanes=16)
    x = torch.zeros(1, 1, 32, 32)
    calls = []

    def fake_forward(self, xx, training, dt, save):
        return torch.zeros(xx.shape[0], 3, xx.shape[2], xx.shape[3]), object()

    def fake_backward(self, sv, g_out, grad_ready=None, allow_plan=True):
        calls.append((grad_ready, allow_plan))
        flat = torch.empty(self.grad_numel)
        views = {}
        for i, (name, p) in enumerate(self.grad_order):
            o = self.grad_offsets[name]
            views[id(p)] = flat[o:o + p.numel()].view(p.shape)
            views[id(p)].fill_(float(i + 1))
        return flat, views

    monkeypatch.setattr(E.Engine, "forward", fake_forward)
    monkeypatch.setattr(E.Engine, "backward", fake_backward)
    # (a) what torch.nn.parallel.replicate does to a replica: parameters become plain non-leaf tensors
    reps = {id(mod): mod._replicate_for_data_parallel() for mod in m.modules()}
    for mod in m.modules():
        r = reps[id(mod)]
        for name, child in mod._modules.items():
            if child is not None:
                r._modules[name] = reps[id(child)]
        for name, p in mod._parameters.items():
            if p is not None:
                r._parameters[name] = p * 2.0            # non-leaf, requires_grad (stands in for Broadcast.apply output)
    replica = reps[id(m)]
    assert not replica.conv11.weight.is_leaf
    with pytest.warns(UserWarning, match="GradAllReducer"):
        out = replica(x)
    out.sum().backward()
    assert calls[-1] == (None, False)                    # no early exchange, no launch plan on this path
    order = {n: i + 1 for i, (n, _) in enumerate(m._grad_completion_order())}
    for n, p in m.named_parameters():                    # the MASTER parameters received the gradients, through the * 2.0
        assert p.grad is not None and torch.equal(p.grad, torch.full_like(p, 2.0 * order[n])), n
    assert "_ubr_flat_grad" not in m.__dict__ and "_ubr_flat_grad" not in replica.__dict__
    # (b) DistributedDataParallel marks its forward through a class attribute; its hooks hang on AccumulateGrad
    m.zero_grad(set_to_none=True)
    from torch.nn.parallel import DistributedDataParallel as DDP
    seen = []
    hooks = [p.register_post_accumulate_grad_hook(lambda t: seen.append(1)) for p in m.parameters()]
    old = DDP._active_ddp_module
    DDP._active_ddp_module = object()
    try:
        out = m(x)
    finally:
        DDP._active_ddp_module = old
    out.sum().backward()
    for h in hooks:
        h.remove()
    assert len(seen) == len(list(m.parameters())), "AccumulateGrad hooks (DDP's reducer) must fire for every parameter"
    for n, p in m.named_parameters():
        assert torch.equal(p.grad, torch.full_like(p, float(order[n]))), n
    # (c) unwrapped: the fast path installs views of the flat buffer and offers the launch plan
    m.zero_grad(set_to_none=True)
    m(x).sum().backward()
    assert calls[-1][1] is True and "_ubr_flat_grad" in m.__dict__
    flat = m.__dict__["_ubr_flat_grad"]
    assert all(p.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for p in m.parameters())
    with pytest.raises(RuntimeError, match="input requires grad"):
        m(x.clone().requires_grad_(True))
    monkeypatch.undo()
    # forward-only use is unaffected by the checks (it still refuses CPU tensors)
    with torch.no_grad(), pytest.raises(RuntimeError, match="ROCm device"):
        m(x)


def test_bench_self_launches_two_ranks_dry_run():
    """`python bench.py --gpus 2` with no launcher environment starts its own ranks (before any GPU call), runs the bucketed
    gradient exchange between them, and prints ONE JSON line from rank 0 -- rehearsed on CPU with gloo (--dry-run: no kernels)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["dry_run"] is True and d["exchange_ok"] is True
    assert d["config"]["global_batch"] == 32 and d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak"


def test_no_16_byte_buffer_store_with_scalar_offset_in_the_built_library():
    """gfx950 + hipcc 7.2: a buffer_store_dwordx3/x4 with an SGPR soffset gets no wait state before a VALU write to its data
    registers, and the store then sends corrupted data (ubresnet_amd/csrc/ubr_conv.hip, buf_store16).  The sources fold scalar
    offsets into the vector offset for such stores; scan the device code of the library build() produced to keep it that way."""
    import importlib.util
    lib = os.path.join(REPO, "ubresnet_amd", "libubresnet_hip.so")
    if not os.path.exists(lib) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("library not built / llvm-objdump not present")
    spec = importlib.util.spec_from_file_location("check_store_hazard", os.path.join(REPO, "tools", "check_store_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    total, bad = mod.scan(lib)
    assert total > 100, "no device code found in %s" % lib
    assert not bad, "16-byte buffer stores with an SGPR soffset: %s" % bad[:4]
