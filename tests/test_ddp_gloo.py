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
        # rank-dependent unused parameters: rank 0 never touches params[1], rank 1 never touches params[3]; the buckets
        # must still be issued in the same order on both ranks (a mismatch pairs buckets of different sizes: hang/corrupt)
        for p in params + [unused]:
            p.grad = None
        skip = 1 if rank == 0 else 3
        loss = sum((rank + 1) * (i + 1) * p.sum() for i, p in enumerate(params) if i != skip)
        loss.backward()
        red.finish()
        for i, p in enumerate(params):
            expect = sum((r + 1) * (i + 1) for r in range(world) if i != (1 if r == 0 else 3)) / world
            assert torch.allclose(p.grad, torch.full_like(p, expect)), ("unused", i, p.grad[:3], expect)
        # a second backward before finish() must raise instead of accumulating into a buffer that is being reduced
        for p in params + [unused]:
            p.grad = None
        sum(p.sum() for p in params).backward()
        try:
            sum(p.sum() for p in params).backward()
            raise AssertionError("second backward before finish() did not raise")
        except RuntimeError as e:
            assert "no_sync" in str(e)
        red.finish()
        # gradient accumulation: first micro-batch under no_sync(), second synchronised
        for p in params + [unused]:
            p.grad = None
        with red.no_sync():
            sum((rank + 1) * p.sum() for p in params).backward()
        sum(2.0 * p.sum() for p in params).backward()
        red.finish()
        for i, p in enumerate(params):
            expect = sum((r + 1) + 2.0 for r in range(world)) / world
            assert torch.allclose(p.grad, torch.full_like(p, expect)), ("accum", i, p.grad[:3], expect)
        # gradient sinks: a backward that asks for the bucket slice of its parameter and produces the gradient there;
        # autograd adopts the slice as .grad, the hook finds it in place and copies nothing
        from vit_amd.ddp import grad_sink

        class ScaleFn(torch.autograd.Function):
            @staticmethod
            def forward(ctx, p, c):
                ctx.c, ctx.ptr = c, p.data_ptr()
                return p.detach() * c

            @staticmethod
            def backward(ctx, g):
                out = grad_sink(ctx.ptr, g.shape, g.dtype)
                assert out is not None
                torch.mul(g, ctx.c, out=out)
                return out, None

        for p in params + [unused]:
            p.grad = None
        target = params[2]
        b, i = red._where[target]
        slot = red._view(b, i)
        (ScaleFn.apply(target, float(rank + 1)).sum() + sum(p.sum() for p in params if p is not target)).backward()
        assert target.grad.data_ptr() == slot.data_ptr()          # adopted, not copied
        assert grad_sink(target.data_ptr(), target.shape, target.dtype) is None   # handed out once per step
        red.finish()
        expect = sum(r + 1 for r in range(world)) / world
        assert torch.allclose(target.grad, torch.full_like(target, expect)), ("sink", target.grad[:3], expect)
        with red.no_sync():
            assert grad_sink(target.data_ptr(), target.shape, target.dtype) is None   # accumulation: no sinks

        class MaybeSinkFn(torch.autograd.Function):   # as the product's backward: the slice if offered, else a new tensor
            @staticmethod
            def forward(ctx, p, c):
                ctx.c, ctx.ptr = c, p.data_ptr()
                return p.detach() * c

            @staticmethod
            def backward(ctx, g):
                out = grad_sink(ctx.ptr, g.shape, g.dtype)
                out = torch.empty_like(g) if out is None else out
                torch.mul(g, ctx.c, out=out)
                return out, None

        # .grad kept between steps and zeroed in place (zero_grad(set_to_none=False)): it still is the bucket slice of the
        # last step, so the slice must not be offered again (the kernel's result would then be added to itself)
        assert target.grad.data_ptr() == slot.data_ptr()
        for p in params + [unused]:
            if p.grad is not None:
                p.grad.zero_()
        (MaybeSinkFn.apply(target, float(rank + 1)).sum() + sum(p.sum() for p in params if p is not target)).backward()
        red.finish()
        assert torch.allclose(target.grad, torch.full_like(target, expect)), ("sink, grad zeroed in place", target.grad[:3], expect)
        # a micro-batch under no_sync(), then a synchronised step that would take the slice: the sum of both, averaged
        for p in params + [unused]:
            p.grad = None
        with red.no_sync():
            MaybeSinkFn.apply(target, 10.0).sum().backward()
        (MaybeSinkFn.apply(target, float(rank + 1)).sum() + sum(p.sum() for p in params if p is not target)).backward()
        red.finish()
        assert torch.allclose(target.grad, torch.full_like(target, 10.0 + expect)), ("sink after no_sync", target.grad[:3])
        red.remove()
        assert grad_sink(target.data_ptr(), target.shape, target.dtype) is None
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback

        q.put((rank, repr(e) + traceback.format_exc()))
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


def _one_rank_worker(port: int, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from vit_amd.ddp import BucketedGradAllReduce

        params = [torch.nn.Parameter(torch.randn(n)) for n in (300, 7, 2000)]
        quiet = BucketedGradAllReduce(params, bucket_bytes=1024 * 4)
        assert not quiet._reduce                       # one rank: no collective unless asked for
        quiet.remove()
        red = BucketedGradAllReduce(params, bucket_bytes=1024 * 4, always_reduce=True)
        assert red._reduce and len(red.buckets) >= 2
        issued = []
        real = dist.all_reduce

        def counting(t, *a, **k):
            issued.append(t.numel())
            return real(t, *a, **k)

        dist.all_reduce = counting
        try:
            for step in range(2):
                for p in params:
                    p.grad = None
                sum((i + 1.5 + step) * p.sum() for i, p in enumerate(params)).backward()
                red.finish()
                for i, p in enumerate(params):   # the average over one rank is the gradient itself
                    assert torch.equal(p.grad, torch.full_like(p, i + 1.5 + step)), (step, i)
            assert len(issued) == 2 * len(red.buckets), issued   # every bucket went through the collective, every step
        finally:
            dist.all_reduce = real
        red.remove()
        q.put("ok")
    except Exception as e:  # noqa: BLE001
        import traceback

        q.put(repr(e) + traceback.format_exc())
    finally:
        dist.destroy_process_group()


def test_always_reduce_runs_the_collectives_in_a_group_of_one():
    """The rehearsal mode of the reducer (tools/rehearse_ddp_one_rank.py runs it on RCCL): a one-rank group still issues
    one all-reduce per bucket and step, and the gradients come back unchanged."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=120)
    p.join(timeout=60)
    assert res == "ok", res


def _encoder_worker(rank: int, world: int, port: int, q):
    """The encoder's own parameter set (prefix tokens, patch embedding, Fourier positions, two blocks, output norm)
    under the reducer, with gradient accumulation: the graph is the CPU oracle's forward over nn.Parameters (the HIP
    path needs a GPU), the reducer and its hooks are the product's."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import vit_oracle as O
        from vit_amd.ddp import BucketedGradAllReduce

        cfg = O.OracleConfig(in_channels=3, patch_size=(8, 8), img_size=(32, 32), depth=2, hidden_size=64,
                             ffn_hidden_size=128, num_attention_heads=2, hidden_dropout=0.0, attention_dropout=0.0,
                             activation="swiglu", pos_enc="fourier", num_register_tokens=2, dtype=torch.float32)
        sd = O.randomize_state(O.init_state(cfg, seed=0), seed=1)
        params = {k: torch.nn.Parameter(v.clone(), requires_grad=v.is_floating_point() and v.numel() > 0 and k != "rope.periods")
                  for k, v in sd.items()}
        x = torch.randn(8, 3, 32, 32, generator=torch.Generator().manual_seed(5))
        red = BucketedGradAllReduce(params.values(), bucket_bytes=16 << 10)
        assert len(red.buckets) >= 4, len(red.buckets)

        def loss_of(batch):
            return O.vit_forward(cfg, params, batch, training=True).float().square().mean()

        mine = x[rank * 4:(rank + 1) * 4]
        for step in range(2):   # second step: re-armed buckets, .grad still pointing into them when set_to_none is False
            for p_ in params.values():
                if step == 0:
                    p_.grad = None
                elif p_.grad is not None:
                    p_.grad.zero_()
            with red.no_sync():
                loss_of(mine[:2]).backward()      # micro-batch 1: accumulates locally, nothing is exchanged
            loss_of(mine[2:]).backward()          # micro-batch 2: hooks fire, buckets go out
            red.finish()
            got = {k: p_.grad.clone() for k, p_ in params.items() if p_.requires_grad}
            # one process, the whole batch: each micro-batch loss is a mean over 2 of the 8 samples, the reducer averages
            # over 2 ranks, so the result is 2 x the gradient of the mean over all 8
            ref = {k: v.detach().clone().requires_grad_(p_.requires_grad) for (k, v), p_ in zip(sd.items(), params.values())}
            O.vit_forward(cfg, ref, x, training=True).float().square().mean().backward()
            for k, g in got.items():
                want = 2.0 * ref[k].grad
                assert torch.allclose(g, want, rtol=1e-4, atol=1e-6 + 1e-4 * want.abs().max().item()), (step, k)
        red.remove()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback

        q.put((rank, repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_encoder_gradient_accumulation_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_encoder_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results
