"""world_size-2 gloo test of the data-parallel gradient exchange (ubresnet_amd/dist.py): buckets are
launched as the executor reports contiguous gradient ranges; the result is the mean over ranks."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ubresnet_amd.dist import GradAllReducer, shard_range

    class M:            # stands in for the model object the executor calls back through
        pass
    m = M()
    red = GradAllReducer(m, bucket_bytes=4 * 1000)
    n = 10000
    flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
    # the executor reports ranges in completion order; emulate 7 uneven stages
    cuts = [0, 123, 1500, 1501, 4000, 7777, 9000, n]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        m._grad_ready_hook(flat, lo, hi)
    red.finish()
    want = torch.arange(n, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
    ok = torch.allclose(flat, want, rtol=1e-6)
    # second step reuses the reducer with a new flat buffer
    flat2 = torch.ones(n) * (rank + 1)
    m._grad_ready_hook(flat2, 0, n)
    red.finish()
    ok = ok and torch.allclose(flat2, torch.full((n,), sum(range(1, world + 1)) / world))
    lo, hi = shard_range(8, rank, world)
    ok = ok and (hi - lo) == 4 and lo == rank * 4
    # construction broadcasts rank 0's parameters and buffers (replicas need not share a seed)
    torch.manual_seed(100 + rank)
    net = torch.nn.Sequential(torch.nn.Conv2d(2, 3, 3), torch.nn.BatchNorm2d(3))
    net[1].running_mean.add_(float(rank))
    red2 = GradAllReducer(net)
    digest = torch.cat([t.detach().reshape(-1).double() for t in list(net.parameters()) + list(net.buffers())])
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    ok = ok and all(torch.equal(gathered[0], t) for t in gathered) and float(net[1].running_mean.abs().max()) == 0.0
    # accumulated passes: .grad tensors that already exist are reduced by finish(), not the in-flight flat buffer
    for p in net.parameters():
        p.grad = torch.full_like(p, float(rank + 1))
    net._grad_accum_pending()
    red2.finish()
    ok = ok and all(torch.allclose(p.grad, torch.full_like(p, sum(range(1, world + 1)) / world)) for p in net.parameters())
    # optional rank-independent BatchNorm statistics for checkpoints
    net[1].running_var.fill_(float(rank))
    red2.average_bn_stats()
    ok = ok and torch.allclose(net[1].running_var, torch.full((3,), (world - 1) / 2.0))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_grad_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
