#!/usr/bin/env python3
"""Headline benchmark: images/sec of one full U-ResNet train step (forward + PixelWiseNLLLoss +
backward + Adam) on synthetic 512x512 LArTPC crops, data-parallel over N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus 8 --steps 20 --warmup 5          (starts its own 8 ranks, see _self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

BASELINE.json configs[1]: ub_uresnet 3-class, bf16 storage / fp32 accumulate, batch 16 per GPU,
512x512x1 -> 3 classes.  Weak scaling: per-GPU batch fixed, global batch = 16*N; the only
exchange is the RCCL gradient all-reduce, overlapped with backward (ubresnet_amd/dist.py).
Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel, timed
live with HIP events on the launch stream) and `cpu_baseline` (the CPU oracle, N=1 only).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
MODEL_BYTES_PER_IMG_BF16 = 1.50e9   # SURVEY.md section 8d "fused-min v1": train step, bf16
MODEL_FLOP_PER_IMG = 199.8e9
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}   # dense peaks, MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (default: a timed region of ~2.5 s)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--inplanes", type=int, default=16)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "f16"])
    ap.add_argument("--model", default="uresnet", choices=["uresnet", "aspp"],
                    help="aspp = BASELINE configs[3]: ASPP_ResNet on 3-plane 512x832 input")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--optimizer", choices=["flat", "torch"], default="flat", help="flat: ubresnet_amd.optim.FlatAdam (one launch); torch: torch.optim.Adam(fused=True)")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--no-infer", action="store_true", help="skip the whole-view inference leg (BASELINE configs[4])")
    ap.add_argument("--breakdown-file", default="")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the ASPP_ResNet (configs[3]) and inplanes=32 (wlarcv2 variant) legs of the N=1 line")
    ap.add_argument("--extra-steps", type=int, default=5, help="timed steps of each extra leg")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the launch path (gloo, no kernels): ranks, gradient exchange and the JSON line; `value` is not a measurement")
    return ap.parse_args()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks ourselves as children of a
    torch.distributed.run subprocess (one process per GPU, rendezvous on 127.0.0.1) BEFORE this process has made any GPU
    call, pass rank 0's JSON line through, exit with the children's status.  (Nothing here touches HIP: the parent only
    imported torch; a process that has initialised the GPU must not exec another program on this pool.)"""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln + "\n")
    if line is not None:
        print(line, flush=True)
    if r.returncode != 0 or line is None:
        raise SystemExit(r.returncode or 1)


def dry_run(a):
    """CPU rehearsal of the N-rank job (tests/test_cpu_host.py): gloo process group, replicated model, bucketed gradient
    exchange through ubresnet_amd.dist.GradAllReducer fed in completion order, barrier + max-over-ranks timing and rank 0's
    JSON line -- everything of bench.py's launch path except the HIP kernels (there is no CPU compute path)."""
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if "RANK" in os.environ:
        dist.init_process_group("gloo")
    from ubresnet_amd.dist import GradAllReducer, shard_range
    from ubresnet_amd.models.ub_uresnet import UResNet
    torch.manual_seed(1234 + rank)             # replicas start DIFFERENT: the reducer's broadcast makes them equal
    model = UResNet(num_classes=3, input_channels=1, inplanes=a.inplanes)
    reducer = GradAllReducer(model, bucket_bytes=8 << 20) if dist.is_initialized() else None
    n = sum(p.numel() for p in model.parameters())
    gb = a.batch * world
    lo, hi = shard_range(gb, rank, world)

    def step(i):
        flat = torch.full((n,), float(rank + 1 + i))
        if reducer is not None:
            for c0 in range(0, n, 3000000):
                model._grad_ready_hook(flat, c0, min(n, c0 + 3000000))
            reducer.finish()
        return flat

    for i in range(a.warmup):
        step(i)
    if dist.is_initialized():
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        flat = step(i)
    if dist.is_initialized():
        dist.barrier()
    el = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    want = sum(r + 1 + a.steps - 1 for r in range(world)) / world
    ok = bool(torch.allclose(flat, torch.full_like(flat, want)))
    digest = torch.cat([p.detach().reshape(-1)[:4] for p in model.parameters()]).double().sum().reshape(1)
    if dist.is_initialized():
        ds = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(ds, digest)
        ok = ok and all(torch.equal(ds[0], d) for d in ds)
    if rank == 0:
        print(json.dumps({"metric": "images/sec, 512x512 U-ResNet train step (fwd+loss+bwd+Adam)", "value": gb * a.steps / el, "unit": "images/sec",
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * el / a.steps, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic", "dry_run": True,
                          "config": {"workload": "DRY RUN on CPU (gloo): launch path and gradient exchange only, no kernels; value is not a measurement",
                                     "parallelism": "dp%d" % world, "global_batch": gb, "shard_of_rank0": [lo, hi]},
                          "exchange_ok": ok}), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


def host_cores():
    """threads the CPU baseline may use: affinity, capped by the cgroup CPU quota of the box"""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("UBR_CPU_THREADS", str(n)))))


def _trained_iou(size, inplanes, steps=200):
    """BASELINE's second metric on weights that mean something: train the HIP path for `steps` bf16 steps on synthetic crops
    (about 3 s), copy the weights to the CPU oracle, and compare class maps on a held-out batch (eval mode, the deployment
    semantics): per-class pixel IoU of the fp32 and bf16 HIP forward against the CPU reference path, plus the trained
    network's own IoU against the labels (so the reader can see it did learn)."""
    from oracle import uresnet_oracle as O
    from ubresnet_amd import synthetic
    from ubresnet_amd.models.ub_uresnet import UResNet
    from ubresnet_amd.optim import FlatAdam
    from ubresnet_amd.training.pixelwise_nllloss import PixelWiseNLLLoss
    torch.manual_seed(99)
    m = UResNet(num_classes=3, input_channels=1, inplanes=inplanes).cuda().train()
    m.compute_dtype = torch.bfloat16
    opt = FlatAdam(m, lr=1e-3, weight_decay=1e-4)
    crit = PixelWiseNLLLoss()
    B = 8
    pool = [tuple(torch.from_numpy(a).cuda() for a in synthetic.make_batch(B, size, size, 7000 + 100 * i)) for i in range(4)]
    first = last = None
    for i in range(steps):
        x, lab, wgt = pool[i % len(pool)]
        loss = crit(m(x), lab, wgt)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if i == 0:
            first = float(loss.detach())
    last = float(loss.detach())
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    xh, labh, _ = synthetic.make_batch(2, size, size, 9000)
    xt = torch.from_numpy(xh)
    with torch.no_grad():
        ref = O.uresnet_forward(sd, xt, False)
        a = ref.argmax(1).reshape(-1)
        top2 = torch.topk(ref, 2, dim=1)[0]
        safe = ((top2[:, 0] - top2[:, 1]) > 0.2).reshape(-1)
        out = {"train_steps": steps, "loss_first": first, "loss_last": last, "pixels": int(a.numel()),
               "margin_gt_0.2_fraction": float(safe.float().mean())}
        lab = torch.from_numpy(labh).reshape(-1)
        cm = torch.bincount(lab * 3 + a, minlength=9).reshape(3, 3)
        out["reference_vs_labels"] = [float(v) for v in O.iou_from_confusion(cm)]
        m.eval()
        for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
            m.compute_dtype = dt
            b = m(xt.cuda()).argmax(1).reshape(-1).cpu()
            for tag, sel in (("", slice(None)), ("_margin_gt_0.2", safe)):
                cm = torch.bincount(a[sel] * 3 + b[sel], minlength=9).reshape(3, 3)
                out[name + tag] = [float(v) for v in O.iou_from_confusion(cm)]
    del m, opt
    return out


def cpu_baseline(size, inplanes, seconds_budget=12.0):
    """The CPU oracle (validated against the reference's own code, tests/test_oracle_golden.py)
    doing the same train step on the host cores: BASELINE config 0 (batch 2, fp32); timed with all host cores the box
    gives this job and with 8 threads (comparable with the survey's 1.32 img/s)."""
    from collections import OrderedDict
    from oracle import uresnet_oracle as O
    from ubresnet_amd import synthetic
    cores = host_cores()
    B = 2
    sd = O.seeded_state_dict(O.uresnet_schema(3, 1, inplanes, 16), 42)
    x, lab, wgt = synthetic.make_batch(B, size, size, 1000)
    xt, lt, wt = torch.from_numpy(x), torch.from_numpy(lab), torch.from_numpy(wgt)
    p = OrderedDict((k, (v.clone().requires_grad_(True) if O.is_param_key(k) else v.clone())) for k, v in sd.items())
    opt = torch.optim.Adam([v for k, v in p.items() if O.is_param_key(k)], lr=1e-5, weight_decay=1e-4)

    def step():
        ns = {}
        logp = O.uresnet_forward(p, xt, True, ns)
        loss = O.pixelwise_nll(logp, lt, wt)
        opt.zero_grad()
        loss.backward()
        opt.step()
        for k, v in ns.items():
            p[k] = v
        return float(loss.detach())

    # BASELINE's second metric: per-class pixel IoU of this build's class map against the CPU reference path, same
    # weights, same batch, train-mode forward (batch statistics), before any update
    iou = None
    try:
        from ubresnet_amd.models.ub_uresnet import UResNet
        torch.set_num_threads(cores)
        with torch.no_grad():
            ref = O.uresnet_forward(p, xt, True, {})
            m = UResNet(num_classes=3, input_channels=1, inplanes=inplanes)
            m.load_state_dict(sd)
            m = m.cuda().train()
            a = ref.argmax(1).reshape(-1)
            top2 = torch.topk(ref, 2, dim=1)[0]
            safe = ((top2[:, 0] - top2[:, 1]) > 0.2).reshape(-1)
            iou = {"weights": "seeded, untrained", "pixels": int(a.numel()), "margin_gt_0.2_fraction": float(safe.float().mean())}
            for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
                m.compute_dtype = dt
                b = m(xt.cuda()).argmax(1).reshape(-1).cpu()
                for tag, sel in (("", slice(None)), ("_margin_gt_0.2", safe)):
                    cm = torch.bincount(a[sel] * 3 + b[sel], minlength=9).reshape(3, 3)
                    iou[name + tag] = [float(v) for v in O.iou_from_confusion(cm)]
            del m
    except Exception as e:     # the IoU report must never take the benchmark line down
        iou = {"error": repr(e)}
    try:
        iou_trained = _trained_iou(size, inplanes)
    except Exception as e:
        iou_trained = {"error": repr(e)}

    def timed(nthreads, budget):
        torch.set_num_threads(nthreads)
        tw = time.time()
        step()
        tw = time.time() - tw
        nmax = max(2, min(12, int(budget / max(tw, 1e-3))))
        ts = []
        for _ in range(nmax):
            t0 = time.time()
            step()
            ts.append(time.time() - t0)
        return nmax, sum(ts) / len(ts), min(ts)

    n, mean_s, min_s = timed(cores, seconds_budget)
    res = {"value": B / mean_s, "unit": "images/sec", "cores": cores, "kind": "port",
           "sample": "%d timed train steps (1 warm-up) of the CPU oracle, UResNet ip%d fp32, batch %d, %dx%d, Adam" % (n, inplanes, B, size, size),
           "ms_per_step": 1e3 * mean_s, "ms_per_step_min": 1e3 * min_s, "iou_vs_reference": iou, "iou_vs_reference_trained": iou_trained}
    if cores != 8:
        n8, mean8, min8 = timed(min(8, cores), seconds_budget / 2)
        res["threads_8"] = {"value": B / mean8, "ms_per_step": 1e3 * mean8, "ms_per_step_min": 1e3 * min8, "steps": n8}
    return res


def infer_leg(events=6, warmup=2):
    """BASELINE.json configs[4] on the same GPU, outside the train timed region: whole-view inference, 3 x 1008 x 3456 event
    = 30 tiles of 512x832, UResNet(ip16, 4 classes; deploy/ubresnet_funcs.py:43), fp16 storage / fp32 accumulate, BatchNorm
    folded, ONE hipGraph replay of all 30 tiles per event (deploy/run_ubresnet_wholeview.py:191-277 shape; tiles are
    independent in eval mode, so the replay batch is a free parameter: 10 -> 30 tiles per replay is +17 % tiles/s)."""
    import numpy as np
    from ubresnet_amd import deploy, synthetic
    torch.manual_seed(7)
    m = deploy.load_model(None, "cuda:%d" % torch.cuda.current_device(), num_classes=4)
    rows, cols = 1008, 3456
    adc = np.zeros((3, 1, rows, cols), np.float32)
    for p in range(3):
        adc[p, 0] = synthetic.make_crop(rows, cols, 5000 + p)[0]
    view = torch.from_numpy(adc).cuda()
    seg = deploy.WholeViewSegmenter(m, rows, cols, planes=3, tile=(512, 832), batch=30, dtype=torch.float16, use_graph=True)
    for _ in range(warmup):
        seg(view)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(events):
        seg(view)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    nt = seg.tiles_per_event
    gb_tile = 0.654e9        # 201.2 M elements per 512^2 image x (512*832 / 512^2) x 2 B
    ach = gb_tile * nt * events / el / 1e9
    # per-kernel breakdown of ONE forward of the 30-tile batch (a timed replay of the recorded launch tape: HIP events around every
    # launch; the timed events above replay the captured hipGraph of the same launches)
    tops = dominant = None
    ktime = None
    try:
        from ubresnet_amd import ops, plan as _plan
        prof = ops.LaunchProfiler()
        ops._prof = prof
        _plan.TIMED = prof.timed if _plan.ENABLED else None
        try:
            with torch.no_grad():
                seg._forward_batch(seg._static_in)
            torch.cuda.synchronize()
        finally:
            ops._prof = None
            _plan.TIMED = None
        dominant, tops, tot, _ = _rooflines(prof, "f16")
        ktime = 1e3 * tot
    except Exception as e:
        tops = {"error": repr(e)}
    return {"roofline_top_kernels": tops, "roofline_dominant_kernel": dominant, "kernel_time_ms_per_event": ktime,
            "metric": "tiles/sec, whole-view 3456x1008 tiled inference (512x832 tiles, forward only, fp16, hipGraph)",
            "value": nt * events / el, "unit": "tiles/sec", "events_per_sec": events / el, "ms_per_event": 1e3 * el / events,
            "tiles_per_event": nt, "events_timed": events, "dtype": "f16", "hipgraph": True, "n_gpus": 1, "data": "synthetic",
            "config": {"workload": "UResNet ip16 4-class eval, 3x1008x3456 views -> 30 tiles of 512x832, one graph replay of 30 tiles"},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_tile": gb_tile, "scope": "whole event (crop + graph replay + stitch)"}}


def _pmc_file():
    """whole-step / per-kernel HBM bytes from the committed rocprofv3 --pmc collection (tools/pmc_traffic.py), only while it
    still describes the kernels being run: the file carries a hash of ubresnet_amd/csrc/ taken when it was collected"""
    try:
        from ubresnet_amd.build import source_hash
        for name in ("r03_pmc_traffic.json", "r02_pmc_traffic.json"):
            f = os.path.join(REPO, "profiles", name)
            if os.path.exists(f):
                pm = json.load(open(f))
                if pm.get("csrc_sha256") == source_hash():
                    return pm
                return None
    except Exception:
        pass
    return None


def _rooflines(prof, dtype, pmc=None):
    """dominant kernel symbol (as rocprofv3 --kernel-trace --stats names it) of a per-launch breakdown taken with HIP events
    on the launch streams, plus the same figures for the next kernels by time"""
    bysym = prof.summary(by="kernel")
    tot = sum(v[1] for v in bysym.values())
    # (launches the operator wrappers do not label -- BatchNorm finalize kernels, memsets -- are one pseudo entry: it
    # counts in the total, it is not a kernel symbol)
    real = {k: v for k, v in bysym.items() if not k.startswith("launches outside")}
    peak_tf = MFMA_PEAK_TFLOPS[dtype]
    ridge = peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9)

    def entry(sy, c_, t_, b_, fl_):
        mf = fl_ / max(b_, 1) > ridge
        ach = fl_ / t_ / 1e12 if mf else b_ / t_ / 1e9
        pk = peak_tf if mf else HBM_PEAK_GBS
        return {"kernel": sy, "launches_per_step": c_, "avg_launch_us": 1e6 * t_ / c_, "bound": "mfma" if mf else "hbm",
                "achieved": ach, "peak": pk, "unit": "TFLOP/s" if mf else "GB/s", "frac": ach / pk, "share_of_gpu_time": t_ / tot}

    sym, (cnt, tsum, nbytes, flops) = max(real.items(), key=lambda kv: kv[1][1])
    top = entry(sym, cnt, tsum, nbytes, flops)
    traffic = None
    if pmc is not None and sym in pmc.get("kernels", {}):
        traffic = pmc["kernels"][sym]["hbm_bytes_per_launch"]
    roof = {"bound": top["bound"], "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"], "frac": top["frac"], "traffic": traffic,
            "kernel": sym, "launches_per_step": cnt, "avg_launch_us": 1e6 * tsum / cnt, "algorithmic_bytes_per_launch": nbytes / cnt,
            "algorithmic_flop_per_launch": flops / cnt, "arithmetic_intensity": flops / max(nbytes, 1), "share_of_gpu_time": tsum / tot}
    tops = [entry(sy, *v) for sy, v in sorted(real.items(), key=lambda kv: -kv[1][1])[:8]]
    return roof, tops, tot, bysym


def train_leg(kind, inplanes, batch, size, dtype, steps, warmup, optimizer, world, rank, dev, breakdown, breakdown_file=""):
    """K timed train steps (forward + PixelWiseNLLLoss + zero_grad + backward + [gradient all-reduce] + Adam) of one model
    configuration, inputs resident in HBM; returns the fields of a bench line for it (rank 0; other ranks get the timing)."""
    from ubresnet_amd import ops, synthetic
    from ubresnet_amd.dist import GradAllReducer, shard_range
    from ubresnet_amd.training.pixelwise_nllloss import PixelWiseNLLLoss
    dt = {"bf16": torch.bfloat16, "f32": torch.float32, "f16": torch.float16}[dtype]
    torch.manual_seed(1234)                     # identical initial weights on every rank (DP replicas)
    if kind == "aspp":
        from ubresnet_amd.models.ASPP_ResNet import ASPP_ResNet
        model = ASPP_ResNet(num_classes=3, in_channels=3, inplanes=16, showsizes=False).to(dev)
        planes, H, W, inplanes = 3, 512, 832, 16
    else:
        from ubresnet_amd.models.ub_uresnet import UResNet
        model = UResNet(num_classes=3, input_channels=1, inplanes=inplanes).to(dev)
        planes, H, W = 1, size, size
    model.compute_dtype = dt
    model.train()
    crit = PixelWiseNLLLoss()
    # reference: Adam(lr 1e-5, weight_decay 1e-4), wlarcv2.py:155-157.  Default: this package's flat Adam (same arithmetic, one
    # launch over the flat parameter / gradient buffers); --optimizer torch uses torch.optim.Adam(fused=True).
    if optimizer == "flat":
        from ubresnet_amd.optim import FlatAdam
        opt = FlatAdam(model, lr=1e-5, weight_decay=1e-4)
    else:
        opt = torch.optim.Adam(list(model.parameters()), lr=1e-5, weight_decay=1e-4, fused=True)
    reducer = GradAllReducer(model, bucket_bytes=int(float(os.environ.get("UBR_BUCKET_MB", "8")) * (1 << 20))) if dist.is_initialized() else None

    # synthetic crops: rank r holds images [r*b, (r+1)*b) of the global batch, resident in HBM
    gb = batch * world
    lo, hi = shard_range(gb, rank, world)
    x, lab, wgt = synthetic.make_batch(batch, H, W, seed0=1000 + lo * planes, planes=planes)
    x, lab, wgt = torch.from_numpy(x).to(dev), torch.from_numpy(lab).to(dev), torch.from_numpy(wgt).to(dev)

    def step():
        out = model.forward(x)
        loss = crit.forward(out, lab, wgt)
        opt.zero_grad()
        loss.backward()
        if reducer is not None:
            reducer.finish()
        opt.step()
        return loss

    for _ in range(warmup):
        step()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    el = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    lossv = float(loss.detach().item())

    name = "ASPP_ResNet" if kind == "aspp" else "ub_uresnet"
    res = {
        "metric": ("images/sec, 512x512 U-ResNet train step (fwd+loss+bwd+Adam)" if kind == "uresnet"
                   else "images/sec, 3x512x832 ASPP-ResNet train step (fwd+loss+bwd+Adam)"),
        "value": gb * steps / el, "unit": "images/sec", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": 1e3 * el / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic",
        "config": {"workload": "%s 3-class ip%d %s, batch %d per GPU (global %d), %dx%dx%d synthetic LArTPC crops, Adam(1e-5, wd 1e-4)"
                               % (name, inplanes, dtype, batch, gb, H, W, planes),
                   "parallelism": "dp%d" % world, "global_batch": gb,
                   "optimizer": "ubresnet_amd.optim.FlatAdam" if optimizer == "flat" else "torch.optim.Adam(fused=True)"},
        "final_loss": lossv,
    }
    esz = 4 if dtype == "f32" else 2
    if kind == "uresnet":   # SURVEY.md section 8d; ip32: 488.9 M elements train-forward, 248.75 GF forward
        if inplanes == 32:
            per_img, flop_img = 3 * 488.9e6 * esz * (size * size / 262144.0), 3 * 248.75e9 * (size * size / 262144.0)
        else:
            per_img = MODEL_BYTES_PER_IMG_BF16 * (esz / 2.0) * (size * size / 262144.0) * (inplanes / 16.0)
            flop_img = MODEL_FLOP_PER_IMG * (size * size / 262144.0) * (inplanes / 16.0) ** 2
    else:   # ASPP ip16 @3x512x832: 442.0 M elements train-forward, 169.77 GF forward
        per_img, flop_img = 3 * 442.0e6 * esz, 3 * 169.77e9
    headline = kind == "uresnet" and inplanes == 16 and size == 512
    pmc = _pmc_file() if headline else None
    if pmc is not None and not (pmc.get("dtype") == dtype and pmc.get("batch") == batch):
        pmc = None
    pmc_step = None
    if pmc is not None:
        pmc_step = pmc.get("hbm_bytes_per_step") or sum(v["launches_profiled"] / 4.0 * v["hbm_bytes_per_launch"] for v in pmc["kernels"].values())
    res["step_model"] = {"pmc_hbm_bytes_per_step": pmc_step, "pmc_over_algorithmic": (pmc_step / (per_img * batch)) if pmc_step else None,
                         "algorithmic_bytes_per_image": per_img, "achieved_GBs_per_gpu": per_img * batch * steps / el / 1e9,
                         "frac_of_hbm_peak": per_img * batch * steps / el / 1e9 / HBM_PEAK_GBS,
                         "achieved_TFLOPs_per_gpu": flop_img * batch * steps / el / 1e12,
                         "frac_of_mfma_peak": flop_img * batch * steps / el / 1e12 / MFMA_PEAK_TFLOPS[dtype]}

    # ---- per-launch breakdown of one more step (HIP events on the launch streams) -> roofline of the dominant kernel
    if breakdown:
        # Launch tapes: the taped launches are timed INSIDE a replay (events around every launch on its own stream, same two-stream
        # overlap as the timed steps and as the rocprofv3 run of this command); the few launches issued from Python (head,
        # loss, stem expansion, optimizer) are timed by the operator wrappers as before.
        from ubresnet_amd import plan as _plan
        prof = ops.LaunchProfiler()
        ops._prof = prof
        _plan.TIMED = prof.timed if _plan.ENABLED else None
        step()                      # every rank runs it: the step contains the gradient all-reduce
        ops._prof = None
        _plan.TIMED = None
        torch.cuda.synchronize()
        if rank == 0:
            roof, tops, tot, bysym = _rooflines(prof, dtype, pmc)
            res["roofline"] = roof
            res["roofline_top_kernels"] = tops
            if headline:
                # the layer round 1's review named (the full-resolution 16->16 3x3 convolutions: 268 MB of algorithmic traffic per launch)
                want = "%dx%dx%dx%d" % (batch, size, size, inplanes)
                agg = [0, 0.0, 0]
                for (nm, sg), (c_, t_, b_, fl_) in prof.summary(by="shape").items():
                    parts = sg.split(" ")
                    if nm == "conv" and "taps9" in parts and parts[0] == want and parts[2] == want and "S2" not in parts:
                        agg[0] += c_; agg[1] += t_; agg[2] += b_
                if agg[0]:
                    res["roofline_fullres_3x3_conv"] = {"layer": "3x3 %d->%d at %dx%d, batch %d (forward and data-gradient launches)" % (inplanes, inplanes, size, size, batch),
                                                        "launches_per_step": agg[0], "avg_launch_us": 1e6 * agg[1] / agg[0], "bound": "hbm", "achieved": agg[2] / agg[1] / 1e9,
                                                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": agg[2] / agg[1] / 1e9 / HBM_PEAK_GBS,
                                                        "algorithmic_bytes_per_launch": agg[2] / agg[0]}
            res["kernel_time_ms_per_step"] = 1e3 * tot
            if breakdown_file:
                with open(breakdown_file, "w") as f:
                    f.write("== by kernel symbol ==\n%-64s %5s %10s %9s %9s %9s %7s\n" % ("kernel", "n", "total_ms", "avg_us", "GB/s", "TFLOP/s", "share"))
                    for sy, (c_, t_, b_, fl_) in sorted(bysym.items(), key=lambda kv: -kv[1][1]):
                        f.write("%-64s %5d %10.3f %9.1f %9.1f %9.1f %6.1f%%\n" % (sy, c_, 1e3 * t_, 1e6 * t_ / c_, b_ / max(t_, 1e-12) / 1e9, fl_ / max(t_, 1e-12) / 1e12, 100 * t_ / tot))
                    f.write("\n== by (op, shape) ==\n%-14s %-62s %5s %10s %9s %9s %7s\n" % ("op", "shape", "n", "total_ms", "GB/s", "TFLOP/s", "share"))
                    for (nm, sg), (c_, t_, b_, fl_) in sorted(prof.summary(by="shape").items(), key=lambda kv: -kv[1][1]):
                        f.write("%-14s %-62s %5d %10.3f %9.1f %9.1f %6.1f%%\n" % (nm, sg, c_, 1e3 * t_, b_ / max(t_, 1e-12) / 1e9, fl_ / max(t_, 1e-12) / 1e12, 100 * t_ / tot))
    del model, opt, reducer, x, lab, wgt
    torch.cuda.empty_cache()
    return res


def main():
    a = parse()
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if a.gpus > 1 and not launched:
        return _self_launch(a)          # before ANY GPU call of this process
    if a.dry_run:
        return dry_run(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or launched:
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local if world > 1 else 0)
    torch.cuda.set_device(dev)

    res = train_leg(a.model, a.inplanes, a.batch, a.size, a.dtype, a.steps, a.warmup, a.optimizer, world, rank, dev,
                    not a.no_breakdown, a.breakdown_file)
    solo = rank == 0 and world == 1
    if solo and not a.no_extra_legs and a.model == "uresnet" and a.inplanes == 16:
        # BASELINE configs[3] (ASPP_ResNet, 3 x 512 x 832; models/ASPP_ResNet.py:291) and the inplanes=32 network that
        # training/train_ubresnet2018_wlarcv2.py:88 builds ("also report ip=32"), on this one GPU, a few steps each
        for key, kind, ip in (("aspp", "aspp", 16), ("ip32", "uresnet", 32)):
            try:
                res[key] = train_leg(kind, ip, a.batch, 512, a.dtype, a.extra_steps, 3, a.optimizer, 1, 0, dev, not a.no_breakdown)
            except Exception as e:          # a secondary leg must never take the headline line down
                res[key] = {"error": repr(e)}
    if solo and not a.no_infer:
        try:
            res["infer"] = infer_leg()
        except Exception as e:
            res["infer"] = {"error": repr(e)}
    if solo and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(a.size, a.inplanes)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
