"""CPU, world_size 2 over gloo: the data-parallel pieces (valor_amd/dist.py).
  * packed all-gather: values/ordering equal to the reference's ddp_allgather_with_grads + ddp_allgather and
    backward returns the LOCAL slice only (utils/distributed.py:62-72);
  * reducer: bucketed all-reduce over the flat gradient arena == sum over ranks, learnt unused-parameter set,
    overlapped launches from the grad hooks on later steps."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    import socket
    with socket.socket() as sk:                # a free port (the reference harness keeps one open in this process on 296xx)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(world, port, fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _gather_case(rank, world):
    from valor_amd.dist import packed_allgather_with_grads
    g = torch.Generator().manual_seed(100 + rank)
    b = 3
    ft = torch.randn(b, 4, 8, generator=g, requires_grad=True)
    fv = torch.randn(b, 2, 8, generator=g, requires_grad=True)
    fa = torch.randn(b, 1, 8, generator=g, requires_grad=True)
    tok = torch.randint(0, 49408, (b, 4), generator=g)
    Ft, Fv, Fa, Tok = packed_allgather_with_grads(ft, fv, fa, tok)
    # a loss every rank computes identically on the gathered tensors (like the contrastive loss)
    w = torch.arange(Ft.numel(), dtype=torch.float32).view_as(Ft)
    loss = (Ft * w).sum() + (Fv ** 2).sum() + Fa.sum()
    loss.backward()
    return dict(Ft=Ft.detach(), Fv=Fv.detach(), Fa=Fa.detach(), Tok=Tok, ft=ft.detach(), fv=fv.detach(), fa=fa.detach(), tok=tok,
                gft=ft.grad, gfv=fv.grad, gfa=fa.grad, w=w)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_packed_allgather_forward_and_local_slice_backward(world):
    r = _run(_gather_case, world)
    for k, kk in (("Ft", "ft"), ("Fv", "fv"), ("Fa", "fa"), ("Tok", "tok")):
        want = torch.cat([r[q][kk] for q in range(world)], dim=0)   # rank-major concat == utils/distributed.py:55-57
        for rank in range(world):
            assert torch.equal(r[rank][k], want), k
    for rank in range(world):
        b = 3
        assert torch.allclose(r[rank]["gft"], r[rank]["w"][rank * b:(rank + 1) * b])       # local slice of dL/dFt
        assert torch.allclose(r[rank]["gfv"], 2 * r[rank]["fv"])
        assert torch.allclose(r[rank]["gfa"], torch.ones_like(r[rank]["fa"]))


def _reducer_case(rank, world):
    from valor_amd.arena import ParamArena
    from valor_amd.dist import Reducer
    entries = [(f"p{i}", (300 + 7 * i,), 0) for i in range(12)]
    arena = ParamArena(entries, torch.float32, "cpu")
    red = Reducer(arena, bucket_bytes=4096)
    assert len(red.buckets) > 2
    outs = []
    for step in range(3):
        arena.grad.zero_()
        red.prepare_backward()
        loss = 0
        for i, (name, p) in enumerate(arena.params.items()):
            if i % 4 == 3:
                continue                                  # unused parameters (find_unused_parameters semantics)
            loss = loss + (p * (rank + 1) * (i + 1 + step)).sum()
        loss.backward()
        active = red.finish_backward()
        outs.append((arena.grad.clone(), sorted(active), len(red.works)))
    return outs


@pytest.mark.parametrize("world", [2, 4, 8])
def test_reducer_sums_arena_and_learns_unused_parameters(world):
    r = _run(_reducer_case, world)
    tri = world * (world + 1) // 2                       # sum over ranks of (rank + 1)
    for step in range(3):
        g0, act0, nworks0 = r[0][step]
        for q in range(1, world):
            gq, actq, _ = r[q][step]
            assert torch.equal(g0, gq) and act0 == actq
        assert all(f"p{i}" not in act0 for i in (3, 7, 11)) and len(act0) == 9
        # expected: d/dp_i sum_ranks (rank+1)*(i+1+step) = world (world + 1) / 2 * (i+1+step)
        from valor_amd.arena import ParamArena
        arena = ParamArena([(f"p{i}", (300 + 7 * i,), 0) for i in range(12)], torch.float32, "cpu")
        for i in range(12):
            o, n, _ = arena.offsets[f"p{i}"]
            want = 0.0 if i % 4 == 3 else float(tri) * (i + 1 + step)
            assert torch.allclose(g0[o:o + n], torch.full((n,), want)), (step, i)
        if step > 0:
            assert nworks0 > 1          # bucketed, hook-launched all-reduces after the first (learning) step


def _reducer_window_case(rank, world):
    """gradient accumulation windows of 3 micro-steps (train_utils.py:311-329). Three ways to reduce the same windows:
    closing = the last micro-step reduces from its hooks, overlapped (what TrainEngine does); flat = every micro-step deferred, the
    window reduced at once behind the last backward (the round-4 behaviour); the second window mixes in a micro-step of ANOTHER
    parameter subset (a task switch inside the window) whose buckets the closing backward never touches."""
    from valor_amd.arena import ParamArena
    from valor_amd.dist import Reducer
    entries = [(f"p{i}", (300 + 7 * i,), 0) for i in range(12)]
    outs = {}
    for how in ("closing", "flat"):
        arena = ParamArena(entries, torch.float32, "cpu")
        red = Reducer(arena, bucket_bytes=4096)

        def micro(used, scale, defer, closing):
            red.prepare_backward(defer=defer, closing=closing)
            loss = 0
            for i, (name, p) in enumerate(arena.params.items()):
                if i in used:
                    loss = loss + (p * (rank + 1) * scale * (i + 1)).sum()
            loss.backward()

        main, other = [i for i in range(12) if i % 4 != 3], [3, 7]
        res = []
        for window in range(3):
            arena.grad.zero_()
            red.reset_task("main")
            for m in range(3):
                last = m == 2
                if window == 1 and m == 0:               # a micro-step of another task inside the window
                    red.reset_task("other")
                    micro(other, 10.0, True, False)
                    assert red.finish_backward(last=False) == set()
                    red.reset_task("main")
                    continue
                micro(main, 1.0 + m, not (last and how == "closing"), last and how == "closing")
                names = red.finish_backward(last=last)
            res.append((arena.grad.clone(), sorted(names), len(red.works)))
        outs[how] = res
    return outs


def test_accumulation_window_reduces_from_the_last_micro_step():
    world = 2
    r = _run(_reducer_window_case, world)
    tri = world * (world + 1) // 2
    from valor_amd.arena import ParamArena
    arena = ParamArena([(f"p{i}", (300 + 7 * i,), 0) for i in range(12)], torch.float32, "cpu")
    for window in range(3):
        gc, nc, wc = r[0]["closing"][window]
        gf, nf, wf = r[0]["flat"][window]
        assert torch.equal(gc, gf) and nc == nf                    # bit-identical to the un-overlapped window
        assert torch.equal(gc, r[1]["closing"][window][0])         # replicas agree
        for i in range(12):
            o, n, _ = arena.offsets[f"p{i}"]
            if i % 4 != 3:
                want = tri * (i + 1) * ((2 + 3) if window == 1 else (1 + 2 + 3))
            else:
                want = tri * (i + 1) * 10.0 if (window == 1 and i in (3, 7)) else 0.0
            assert torch.allclose(gc[o:o + n], torch.full((n,), float(want))), (window, i)
        if window == 1:
            assert "p3" in nc and "p7" in nc and "p11" not in nc
        if window > 0:
            assert wc > 1            # the learnt task: bucket by bucket from the closing backward's hooks


def _reducer_multiuse_case(rank, world):
    """gradients written straight into the arena by kernels (ops.GradSink.listener), some parameters TWICE per backward (the
    shared BERT of the VideoSwin variant: text-encoder pass + decoder passes): a bucket may only fly after the last write."""
    from valor_amd import ops
    from valor_amd.arena import ParamArena
    from valor_amd.dist import Reducer
    entries = [(f"p{i}", (500 + 3 * i,), 0) for i in range(10)]
    arena = ParamArena(entries, torch.float32, "cpu")
    red = Reducer(arena, bucket_bytes=4096)
    shared = {"p2", "p3", "p6"}
    outs = []
    try:
        for step in range(3):
            arena.grad.zero_()
            red.prepare_backward()
            names = list(arena.params)
            for n in reversed(names):                     # "decoder passes": every parameter once
                arena.params[n].grad += float(rank + 1)
                ops.GradSink.listener(n)
            for n in reversed(names):                     # "text-encoder pass": the shared ones again, later
                if n in shared:
                    arena.params[n].grad += 10.0 * (rank + 1) * (step + 1)
                    ops.GradSink.listener(n)
            red.finish_backward()
            outs.append((arena.grad.clone(), dict(red.uses), len(red.works)))
    finally:
        ops.GradSink.listener = None
    return outs


def test_reducer_waits_for_every_write_of_a_shared_parameter():
    r = _run(_reducer_multiuse_case)
    from valor_amd.arena import ParamArena
    arena = ParamArena([(f"p{i}", (500 + 3 * i,), 0) for i in range(10)], torch.float32, "cpu")
    for step in range(3):
        g0, uses, nworks = r[0][step]
        assert torch.equal(g0, r[1][step][0])
        assert uses["p2"] == 2 and uses["p0"] == 1
        for i in range(10):
            o, n, _ = arena.offsets[f"p{i}"]
            want = 3.0 + (30.0 * (step + 1) if i in (2, 3, 6) else 0.0)        # sum over ranks 1 + 2 (+ 10 + 20 per step)
            assert torch.allclose(g0[o:o + n], torch.full((n,), want)), (step, i, float(g0[o]))
        if step > 0:
            assert nworks > 1


def _reducer_fp32_mode_case(rank, world):
    """mode "fp32": a bf16 gradient arena summed across ranks in fp32 and rounded once (gloo, CPU)"""
    from valor_amd.arena import ParamArena
    from valor_amd.dist import Reducer
    entries = [(f"p{i}", (2048,), 0) for i in range(4)]
    out = {}
    for mode in ("allreduce", "fp32"):
        arena = ParamArena(entries, torch.bfloat16, "cpu")
        red = Reducer(arena, bucket_bytes=4096, mode=mode)
        g = torch.Generator().manual_seed(7 + rank)
        for step in range(2):
            arena.grad.zero_()
            red.prepare_backward()
            for n, p in arena.params.items():
                p.grad.copy_(torch.randn(p.shape, generator=g).bfloat16())
                red._on_grad(n)
            red.finish_backward()
        out[mode] = arena.grad.float().clone()
    return out


def test_reducer_fp32_accumulation_mode():
    r = _run(_reducer_fp32_mode_case)
    # what every rank fed in, exactly
    parts = []
    for rank in range(2):
        g = torch.Generator().manual_seed(7 + rank)
        last = None
        for step in range(2):
            last = torch.cat([torch.randn((2048,), generator=g).bfloat16().float() for _ in range(4)])
        parts.append(last)
    exact = parts[0] + parts[1]
    for mode in ("allreduce", "fp32"):
        assert torch.equal(r[0][mode], r[1][mode])
        assert torch.equal(r[0][mode], exact.bfloat16().float()), mode      # two addends: one rounding either way


def test_bf16_cross_rank_sum_error():
    """What summing the gradient arena across 8 ranks IN bf16 (ring order: 7 roundings per element) costs against an fp32 sum rounded
    once (Reducer mode "fp32"): per element ~0.3-0.5 % rms -- the size of the bf16 rounding every gradient already carries -- and
    < 1e-4 on the global norm the clipping uses (the north-star's gradient checks are at 2e-3). Pure arithmetic, no process group."""
    g = torch.Generator().manual_seed(0)
    n, world = 1 << 20, 8
    scale = torch.exp(torch.randn(n, generator=g))                       # gradients spanning several octaves
    ranks = [(torch.randn(n, generator=g) * scale).bfloat16() for _ in range(world)]
    exact = sum(r.double() for r in ranks)
    acc = ranks[0].clone()
    for r in ranks[1:]:
        acc = (acc.float() + r.float()).bfloat16()                        # what a bf16 ring reduction does at every hop
    once = sum(r.float() for r in ranks).bfloat16()
    rel = lambda a: float((a.double() - exact).norm() / exact.norm())
    nrm = lambda a: abs(float(a.double().norm() / exact.norm()) - 1.0)
    assert rel(once) < 2.5e-3 and rel(acc) < 6e-3 and rel(acc) > rel(once)
    assert nrm(acc) < 1e-4 and nrm(once) < 1e-4


def _unequal_gather_case(rank, world):
    from valor_amd.dist import all_gather_list, ddp_allgather
    g = torch.Generator().manual_seed(300 + rank)
    n = 3 + 2 * rank                                      # 3, 5, 7 ... rows: the last validation batches differ between ranks
    x = torch.randn(n, 4, 6, generator=g)
    tok = torch.randint(0, 30522, (n, 9), generator=g)
    return dict(x=x, tok=tok, X=ddp_allgather(x), T=ddp_allgather(tok), objs=all_gather_list({"rank": rank, "ids": [f"v{rank}_{i}" for i in range(n)]}))


@pytest.mark.parametrize("world", [2, 3])
def test_ddp_allgather_with_unequal_sizes(world):
    """utils/distributed.py:77-93 (ddp_allgather: sizes, pad to the largest, gather, cut) and :133-170 (all_gather_list), as
    test.py:275-290 uses them on the validation features: every rank receives the ranks' rows in rank order, padding removed"""
    res = _run(_unequal_gather_case, world)
    want_x, want_t = torch.cat([r["x"] for r in res]), torch.cat([r["tok"] for r in res])
    for r in res:
        assert torch.equal(r["X"], want_x) and torch.equal(r["T"], want_t)
        assert [o["rank"] for o in r["objs"]] == list(range(world))
        assert [i for o in r["objs"] for i in o["ids"]] == [f"v{k}_{i}" for k in range(world) for i in range(3 + 2 * k)]
