"""Launch plans (ubresnet_amd/plan.py, ubr_tape_* of include/ubresnet_hip.h): a replayed pass must be bit-identical to the
Python-scheduled pass it was recorded from -- same kernels, same arguments, same stream structure -- for the train step
(forward, two-stream backward, optimizer) and for eval inference; and it must step aside whenever replay is not safe."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import uresnet_oracle as O
from ubresnet_amd import synthetic

if torch.cuda.is_available():
    from ubresnet_amd import plan
    from ubresnet_amd.models.ub_uresnet import UResNet
    from ubresnet_amd.models.ASPP_ResNet import ASPP_ResNet
    from ubresnet_amd.optim import FlatAdam
    from ubresnet_amd.training.pixelwise_nllloss import PixelWiseNLLLoss


def _make(kind, dt):
    if kind == "uresnet":
        sd = O.seeded_state_dict(O.uresnet_schema(3, 1, 16, 16), 42)
        m = UResNet(3, 1, 16)
        planes = 1
    else:
        sd = O.seeded_state_dict(O.aspp_resnet_schema(3, 3, 16), 44)
        m = ASPP_ResNet(3, 3, 16, False)
        planes = 3
    m.load_state_dict(sd)
    m = m.cuda().train()
    m.compute_dtype = dt
    return m, planes


def _train(kind, dt, steps, enabled, monkeypatch):
    monkeypatch.setattr(plan, "ENABLED", enabled)
    m, planes = _make(kind, dt)
    opt = FlatAdam(m, lr=1e-3, weight_decay=1e-4)
    crit = PixelWiseNLLLoss()
    outs, losses = [], []
    for i in range(steps):
        x, lab, wgt = synthetic.make_batch(2, 64, 96, 1000 + 10 * i, planes=planes)
        out = m(torch.from_numpy(x).cuda())
        loss = crit(out, torch.from_numpy(lab).cuda(), torch.from_numpy(wgt).cuda())
        opt.zero_grad()
        loss.backward()
        opt.step()
        outs.append(out.detach().clone())
        losses.append(float(loss))
    torch.cuda.synchronize()
    eng = m.__dict__["_ubr_engine"]
    bufs = {k: v.clone() for k, v in m.state_dict().items()}
    return outs, losses, opt.flat.clone(), bufs, eng


@pytest.mark.parametrize("kind,dt", [("uresnet", torch.float32), ("uresnet", torch.bfloat16), ("aspp", torch.bfloat16)])
def test_replayed_train_steps_equal_python_scheduled_steps(kind, dt, monkeypatch):
    o1, l1, p1, b1, eng1 = _train(kind, dt, 4, True, monkeypatch)
    o0, l0, p0, b0, eng0 = _train(kind, dt, 4, False, monkeypatch)
    assert len(eng1._planned) == 1 and not eng0._planned
    pl = next(iter(eng1._planned.values()))
    assert pl.uses == 4 and pl.bwd is not None and pl.fwd.tape.size() > 100 and pl.bwd.tape.size() > 200
    assert l1 == l0
    for a, b in zip(o1, o0):
        assert torch.equal(a, b)
    assert torch.equal(p1, p0), "parameters differ after 4 steps (1 recorded + 3 replayed)"
    for k in b0:
        assert torch.equal(b1[k], b0[k]), k            # incl. BatchNorm running statistics and num_batches_tracked


def test_outputs_are_fresh_tensors_and_inputs_are_read_in_place(monkeypatch):
    monkeypatch.setattr(plan, "ENABLED", True)
    m, _ = _make("uresnet", torch.float32)
    crit = PixelWiseNLLLoss()
    xs = [torch.from_numpy(synthetic.make_batch(1, 64, 64, 1000 + i)[0]).cuda() for i in range(3)]
    lab = torch.zeros((1, 64, 64), dtype=torch.int64, device="cuda")
    wgt = torch.ones((1, 64, 64), device="cuda")
    outs = []
    for x in xs:
        out = m(x)
        crit(out, lab, wgt).backward()
        m.zero_grad()
        outs.append(out)
    assert len({o.data_ptr() for o in outs}) == 3
    assert not torch.equal(outs[0], outs[1])            # earlier outputs were not overwritten by later passes
    m.eval()
    with torch.no_grad():
        e = [m(x) for x in xs]
    monkeypatch.setattr(plan, "ENABLED", False)
    with torch.no_grad():
        for x, got in zip(xs, e):
            assert torch.equal(m(x), got)


def test_plan_steps_aside_when_replay_is_not_safe(monkeypatch):
    monkeypatch.setattr(plan, "ENABLED", True)
    m, _ = _make("uresnet", torch.float32)
    crit = PixelWiseNLLLoss()
    x, lab, wgt = [torch.from_numpy(a).cuda() for a in synthetic.make_batch(1, 64, 64, 1000)]
    x2 = torch.from_numpy(synthetic.make_batch(1, 64, 64, 2000)[0]).cuda()

    def grads():
        return torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()

    crit(m(x), lab, wgt).backward(); ga = grads(); m.zero_grad()           # records
    crit(m(x2), lab, wgt).backward(); gb = grads(); m.zero_grad()          # replays
    # two forwards before either backward: the second one must not reuse the buffers the first one saved
    oa, ob = m(x), m(x2)
    crit(ob, lab, wgt).backward(); g2 = grads(); m.zero_grad()
    crit(oa, lab, wgt).backward(); g1 = grads(); m.zero_grad()
    assert torch.equal(g1, ga) and torch.equal(g2, gb)
    # accumulation into existing .grad tensors (which alias the plan's flat buffer) must add, not overwrite
    crit(m(x), lab, wgt).backward()
    crit(m(x2), lab, wgt).backward()
    assert torch.allclose(grads(), ga + gb, rtol=1e-5, atol=1e-8)
    m.zero_grad()
    # replaced parameter storage invalidates the baked addresses: the plan is rebuilt, results unchanged
    eng = m.__dict__["_ubr_engine"]
    before = next(iter(eng._planned.values()))
    with torch.no_grad():
        m.conv11.weight.data = m.conv11.weight.data.clone()
    crit(m(x), lab, wgt).backward()
    assert torch.equal(grads(), ga)
    assert next(iter(eng._planned.values())) is not before


def test_dropped_pass_and_changed_batchnorm_scalars(monkeypatch):
    """(a) a forward with gradients enabled whose loss is never back-propagated (the reference's validate() does exactly that,
    training/train_ubresnet2018_wlarcv2.py:428) must not park the plan "in flight" for ever: the next train step replays;
    (b) BatchNorm momentum / eps are baked into a tape at record time, so changing them retires the plan (the eager path
    honours them at once); momentum=None is cumulative averaging as in nn.BatchNorm2d."""
    monkeypatch.setattr(plan, "ENABLED", True)
    m, _ = _make("uresnet", torch.float32)
    crit = PixelWiseNLLLoss()
    x, lab, wgt = [torch.from_numpy(a).cuda() for a in synthetic.make_batch(1, 64, 64, 1000)]
    crit(m(x), lab, wgt).backward(); m.zero_grad()          # records forward and backward
    crit(m(x), lab, wgt).backward(); m.zero_grad()          # replays
    eng = m.__dict__["_ubr_engine"]
    pl = next(iter(eng._planned.values()))
    uses = pl.uses
    out = m(x)                                               # gradients enabled, node dropped without a backward
    assert pl.in_flight
    del out
    out2 = m(x)                                              # the plan is reclaimed, not bypassed
    assert pl.uses == uses + 2 and pl.in_flight
    crit(out2, lab, wgt).backward()
    assert not pl.in_flight
    g_ref = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
    m.zero_grad()
    # (b) momentum 0: running statistics must stop moving although the recorded tape used 0.1
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.momentum = 0.0
    rm = m.bn10.running_mean.clone()
    crit(m(x), lab, wgt).backward()
    torch.cuda.synchronize()
    assert torch.equal(m.bn10.running_mean, rm), "a replayed tape ignored the new BatchNorm momentum"
    assert next(iter(eng._planned.values())) is not pl
    assert torch.equal(torch.cat([p.grad.reshape(-1) for p in m.parameters()]), g_ref)
    m.zero_grad()
    # momentum=None: cumulative average over the batches seen since the reset
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.reset_running_stats()
            mod.momentum = None
    xs = [torch.from_numpy(synthetic.make_batch(1, 64, 64, 3000 + 7 * i)[0]).cuda() for i in range(3)]
    means = []
    for xi in xs:
        with torch.no_grad():
            m(xi)
        means.append(m.bn1.running_mean.clone())
    assert int(m.bn1.num_batches_tracked) == 3
    ref = torch.nn.BatchNorm2d(16, momentum=None).cuda().train()
    conv1 = torch.nn.Conv2d(1, 16, 7, 1, 3).cuda()
    conv1.load_state_dict({"weight": m.conv1.weight.detach(), "bias": m.conv1.bias.detach()})
    with torch.no_grad():
        for xi, got in zip(xs, means):
            ref(conv1(xi))
            assert torch.allclose(got, ref.running_mean, rtol=1e-4, atol=1e-5)
    assert torch.allclose(m.bn1.running_var, ref.running_var, rtol=1e-3, atol=1e-5)


def test_timed_replay_times_every_operator_and_changes_nothing(monkeypatch):
    """bench.py's breakdown: with plan.TIMED set, a replay brackets every taped launch with timing events on its own stream
    (ubr_tape_replay_timed) and reports one record per labelled operator call; the results of the step are those of an
    ordinary replay."""
    from ubresnet_amd import ops
    monkeypatch.setattr(plan, "ENABLED", True)
    m, planes = _make("uresnet", torch.bfloat16)
    crit = PixelWiseNLLLoss()
    x, lab, wgt = synthetic.make_batch(2, 64, 96, 1234, planes=planes)
    xt, lt, wt = torch.from_numpy(x).cuda(), torch.from_numpy(lab).cuda(), torch.from_numpy(wgt).cuda()

    def step():
        m.zero_grad()
        out = m(xt)
        crit(out, lt, wt).backward()
        torch.cuda.synchronize()
        return out.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters()}

    m.bn1.momentum = 0.0
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.momentum = 0.0          # (running statistics do not enter train-mode outputs; keep the steps identical anyway)
    step()                               # records
    o_ref, g_ref = step()                # ordinary replay
    prof = ops.LaunchProfiler()
    monkeypatch.setattr(ops, "_prof", prof)
    monkeypatch.setattr(plan, "TIMED", prof.timed)
    o_t, g_t = step()
    monkeypatch.setattr(ops, "_prof", None)
    monkeypatch.setattr(plan, "TIMED", None)
    assert torch.equal(o_ref, o_t)
    for n in g_ref:
        assert torch.equal(g_ref[n], g_t[n]), n
    names = {r[0] for r in prof.timed}
    assert {"conv", "wgrad", "wgrad_reduce", "block_tail_fwd", "block_tail_bwd_apply", "bn_bwd_reduce", "maxpool_fwd"} <= names, names
    assert len(prof.timed) > 150 and all(r[5] > 0.0 for r in prof.timed)
    by_kernel = prof.summary(by="kernel")
    assert any(k.startswith("wgrad_kernel<bf16_t") for k in by_kernel) and any(k.startswith("conv_") for k in by_kernel)
    o_after, _ = step()                  # and ordinary replays keep working afterwards
    assert torch.equal(o_ref, o_after)
