"""World-size-2 CPU (gloo) tests of the data-parallel host logic: task sharding, flat
parameter buffer, bucketed gradient all-reduce and its equivalence with the global-batch
gradient.  No GPU work: gradients come from a plain torch loss on the CPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from npf_gwwaveform_amd.parallel import BucketedGradReducer, FlatParameters, shard_range


def test_shard_range_partitions_tasks():
    for n, w in [(256, 8), (10, 4), (3, 8), (8192, 8)]:
        cover = []
        for r in range(w):
            a, b = shard_range(n, r, w)
            cover += list(range(a, b))
        assert cover == list(range(n))
        sizes = [shard_range(n, r, w)[1] - shard_range(n, r, w)[0] for r in range(w)]
        assert max(sizes) - min(sizes) <= 1


def _tiny_model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(3, 8), torch.nn.ReLU(), torch.nn.Linear(8, 8), torch.nn.ReLU(),
                               torch.nn.Linear(8, 2))


def test_flat_parameters_are_views_and_single_process_reducer_copies_grads():
    m = _tiny_model()
    ref = [p.detach().clone() for p in m.parameters()]
    flat = FlatParameters(m.parameters())
    for p, r in zip(m.parameters(), ref):
        assert torch.equal(p, r)
        assert p.data_ptr() >= flat.flat.data_ptr()
    red = BucketedGradReducer(flat, world=1, bucket_bytes=64)
    assert len(red.buckets) > 1
    x = torch.randn(5, 3)
    m(x).square().sum().backward()
    g = red.finish()
    want = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    assert torch.equal(g, want)
    with torch.no_grad():
        flat.flat.data.add_(1.0)
    assert torch.allclose(next(m.parameters()), ref[0] + 1.0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    m = _tiny_model()
    flat = FlatParameters(m.parameters())
    red = BucketedGradReducer(flat, bucket_bytes=128)
    g = torch.Generator().manual_seed(42)
    X, Y = torch.randn(8, 3, generator=g), torch.randn(8, 2, generator=g)
    a, b = shard_range(8, rank, world)
    out = []
    for step in range(2):  # two steps: reset() must re-arm the buckets
        for p in flat.params:
            p.grad = None
        red.reset()
        ((m(X[a:b]) - Y[a:b]).square().sum(1)).mean(0).backward()  # local mean over local tasks
        out.append(red.finish().clone())
    q.put((rank, out[0].tolist(), out[1].tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo_allreduce_equals_global_batch_gradient():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=90) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    # single-process global-batch gradient
    m = _tiny_model()
    g = torch.Generator().manual_seed(42)
    X, Y = torch.randn(8, 3, generator=g), torch.randn(8, 2, generator=g)
    ((m(X) - Y).square().sum(1)).mean(0).backward()
    want = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    for rank, g0, g1 in res:
        torch.testing.assert_close(torch.tensor(g0), want, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(torch.tensor(g1), want, rtol=1e-6, atol=1e-7)
