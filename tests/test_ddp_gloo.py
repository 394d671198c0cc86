"""World-size-2 gloo test (CPU) of the bucketed gradient all-reduce used by the data-parallel path."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vit_amd.ddp import BucketedGradAllReduce

        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.randn(n)) for n in (1000, 37, 5000, 64, 3)]
        frozen = torch.nn.Parameter(torch.randn(5), requires_grad=False)
        unused = torch.nn.Parameter(torch.randn(11))
        red = BucketedGradAllReduce(params + [frozen, unused], bucket_bytes=4096 * 4)
        assert len(red.buckets) >= 3
        for step in range(2):  # second step checks re-arming and grads living in the flat buffers
            for p in params + [unused]:
                p.grad = None
            loss = sum(((rank + 1 + step) * (i + 1)) * p.sum() for i, p in enumerate(params))
            loss.backward()
            red.finish()
            for i, p in enumerate(params):
                expect = sum((r + 1 + step) * (i + 1) for r in range(world)) / world
                assert torch.allclose(p.grad, torch.full_like(p, expect)), (step, i, p.grad[:3], expect)
            assert torch.all(unused.grad == 0)
            assert frozen.grad is None
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results


def test_single_process_is_a_no_op():
    from vit_amd.ddp import BucketedGradAllReduce

    p = torch.nn.Parameter(torch.ones(10))
    red = BucketedGradAllReduce([p])
    (p * 3).sum().backward()
    red.finish()
    assert torch.all(p.grad == 3)
