"""CPU, world_size 2, gloo: the gradient-exchange logic (bucket planning in backward order, mean
reduction, per-edge readiness) produces on every rank exactly what the reference's
Accumulate + Broadcast produces: grad = sum over ranks / num_processes (src/convnet.cc:407-450)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from convnet_amd.data_parallel import FlatExchange, GradientExchange, plan_buckets


def test_plan_buckets_orders_and_sizes():
    slices = [("out", 900, 100), ("fc7", 600, 300), ("fc6", 300, 300), ("c2", 100, 50), ("c1", 0, 10)]
    b = plan_buckets(slices, bucket_bytes=4 * 250)
    assert b == [["out", "fc7"], ["fc6"], ["c2", "c1"]]
    assert plan_buckets(slices, 1) == [[k] for k, _, _ in slices]
    assert plan_buckets(slices, 1 << 30) == [[k for k, _, _ in slices]]
    # the tail rule GradientExchange uses: the last slice (the first layer's gradient, final only when the step ends) travels
    # alone, so what shared its bucket goes out one layer earlier
    assert plan_buckets(slices, 4 * 250, split_tail=True) == [["out", "fc7"], ["fc6"], ["c2"], ["c1"]]
    assert plan_buckets(slices, 1 << 30, split_tail=True) == [["out", "fc7", "fc6", "c2"], ["c1"]]
    assert plan_buckets(slices, 1, split_tail=True) == [[k] for k, _, _ in slices]
    assert plan_buckets(slices[:1], 1 << 30, split_tail=True) == [["out"]]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeEdge:
    def __init__(self, name, tied_to=None):
        self.name = name
        self.tied_edge_ = tied_to
        self.num_shares_ = 1
        if tied_to is not None:
            tied_to.num_shares_ += 1

    def GetName(self):
        return self.name

    def IsBackPropBlocked(self):
        return False

    def IsTied(self):
        return self.tied_edge_ is not None


class _FakeLayer:
    def __init__(self, edges):
        self.outgoing_edge_ = edges


class _FakeFlat:
    def __init__(self, t):
        self.t = t

    def tensor(self):
        return self.t

    def GetNumEls(self):
        return self.t.numel()


class _FakeNet:
    """Three weighted edges in a chain a->b->c->d, parameter order == model order (src/convnet.cc:286-298)."""

    def __init__(self, flat):
        self.e = [_FakeEdge("a:b"), _FakeEdge("b:c"), _FakeEdge("c:d")]
        self.layers_ = [_FakeLayer([self.e[0]]), _FakeLayer([self.e[1]]), _FakeLayer([self.e[2]]), _FakeLayer([])]
        self.edge_slices_ = {self.e[0]: (0, 200), self.e[1]: (256, 1000), self.e[2]: (1280, 130)}
        self.grad_parameters_ = _FakeFlat(flat)


def _worker(rank, world, port, bucket_bytes, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = 1280 + 256
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(total, generator=g)
    local = flat.clone()
    net = _FakeNet(flat)
    ex = GradientExchange(bucket_bytes=bucket_bytes, overlap=False)
    ex.Register(net)
    ex.StartStep()
    with pytest.raises(RuntimeError):
        ex.WaitFor(net.e[2])              # nothing exchanged yet
    for e in reversed(net.e):             # backward order: c:d first
        ex.GradReady(e)
    for e in net.e:
        ex.WaitFor(e)
    # reference semantics: every rank ends with sum/num_processes
    gathered = [torch.zeros(total) for _ in range(world)]
    dist.all_gather(gathered, local)
    want = sum(gathered) / world
    ok = True
    for e, (o, n) in net.edge_slices_.items():
        ok &= torch.allclose(flat[o:o + n], want[o:o + n], atol=1e-6)
    # and replicas are bit-identical
    mine = [torch.zeros(total) for _ in range(world)]
    dist.all_gather(mine, flat)
    ok &= all(torch.equal(mine[0], m) for m in mine)
    # FlatExchange (same planner, keys instead of edges)
    flat2 = local.clone()
    fx = FlatExchange([("c:d", 1280, 130), ("b:c", 256, 1000), ("a:b", 0, 200)], bucket_bytes)
    fx.StartStep()
    for k in ("c:d", "b:c", "a:b"):
        fx.GradReady(flat2, k)
    for k, (o, n) in (("a:b", (0, 200)), ("b:c", (256, 1000)), ("c:d", (1280, 130))):
        fx.WaitFor(k)
        ok &= torch.allclose(flat2[o:o + n], want[o:o + n], atol=1e-6)
    out[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes", [1, 4 * 1100, 1 << 30])
def test_gradient_exchange_two_ranks_gloo(bucket_bytes):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, bucket_bytes, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_register_places_a_tied_owner_where_its_last_sharer_completes():
    """an edge tied to another one owns no slice; the owner's slice is final only after the LAST sharer's
    ComputeOuter in backward order — the owner must be bucketed there, not at its own position."""
    class Net:
        pass
    owner, mid = _FakeEdge("a:b"), _FakeEdge("b:c")
    tied = _FakeEdge("c:d", tied_to=owner)          # nearer the output: runs FIRST in backward order
    net = Net()
    net.layers_ = [_FakeLayer([owner]), _FakeLayer([mid]), _FakeLayer([tied]), _FakeLayer([])]
    net.edge_slices_ = {owner: (0, 200), mid: (256, 1000)}
    net.grad_parameters_ = _FakeFlat(torch.zeros(1280))
    ex = GradientExchange.__new__(GradientExchange)
    ex.bucket_bytes_, ex.comm_stream_, ex.transport_ = 1, None, "torch"
    ex.Register(net)
    assert ex.buckets_ == [[mid], [owner]]
    # owner nearer the output than its sharer: final at the sharer's position (last in backward order)
    owner2, mid2 = _FakeEdge("c:d"), _FakeEdge("b:c")
    tied2 = _FakeEdge("a:b", tied_to=owner2)
    net.layers_ = [_FakeLayer([tied2]), _FakeLayer([mid2]), _FakeLayer([owner2]), _FakeLayer([])]
    net.edge_slices_ = {owner2: (1280, 130), mid2: (256, 1000)}
    ex.Register(net)
    assert ex.buckets_ == [[mid2], [owner2]]


def _worker_twice(rank, world, port, out):
    """bench.py's order of events on more than one rank: a weak-scaling run with one exchange, then — same process, same process group —
    the `strong` leg builds a second net and a SECOND exchange after closing the first."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = 1280 + 256
    ok = True
    for round_, bucket_bytes in enumerate((4 * 1100, 1 << 30)):
        g = torch.Generator().manual_seed(1000 * round_ + 100 + rank)
        flat = torch.randn(total, generator=g)
        local = flat.clone()
        net = _FakeNet(flat)
        ex = GradientExchange(bucket_bytes=bucket_bytes, overlap=False)
        ex.Register(net)
        for _step in range(2):
            ex.StartStep()
            for e in reversed(net.e):
                ex.GradReady(e)
            for e in net.e:
                ex.WaitFor(e)
        gathered = [torch.zeros(total) for _ in range(world)]
        dist.all_gather(gathered, local)
        want = sum(gathered) / world          # second step averages already-equal replicas: unchanged
        for e, (o, n) in net.edge_slices_.items():
            ok &= torch.allclose(flat[o:o + n], want[o:o + n], atol=1e-6)
        ok &= ex.SumScalars([float(rank + 1), 2.0]) == [float(sum(range(1, world + 1))), 2.0 * world]
        ex.Close()
        ex.Close()                              # idempotent
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_two_exchanges_one_after_the_other_in_one_process_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_twice, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}
