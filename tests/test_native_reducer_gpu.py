"""The library's own gradient reducer (csrc/reducer.hip, valor_reducer_* of include/valor_hip.h) over RCCL on the GPU box. One rank is all
a 1-GPU box offers: the communicator, the communication stream, the per-bucket events and the in-place collectives all execute, the sum over
one rank is the input, and the stream-side ordering is checked with gradient writes still in flight on two compute streams. The Python
reducer drives it exactly as it would with N ranks (dist.Reducer(native=True))."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, ROOT)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    try:
        from valor_amd import streams
        from valor_amd.arena import ParamArena
        from valor_amd.dist import Reducer
        res = {}
        for mode in ("allreduce", "rs_ag"):
            arena = ParamArena([(f"p{i}", (50000 + 8 * i,), 0) for i in range(6)], torch.bfloat16, "cuda:0")
            red = Reducer(arena, bucket_bytes=65536, mode=mode, native=True)
            res[mode + "_created"] = red.native is not None and len(red.buckets) > 1
            g = torch.Generator().manual_seed(3)
            want = torch.randn(arena.numel, generator=g).bfloat16().cuda()
            # gradient writes in flight on the main AND the side stream when the buckets are launched: a long kernel chain, then the copy
            streams.set_main("cuda:0")
            side = streams.side_stream("cuda:0")
            busy = torch.randn((4096, 4096), device="cuda:0")
            half = arena.numel // 2
            for _ in range(20):
                busy = busy @ busy * 1e-3
            arena.grad[:half].copy_(want[:half])
            with torch.cuda.stream(side):
                side.wait_stream(torch.cuda.current_stream())
                b2 = busy
                for _ in range(20):
                    b2 = b2 @ b2 * 1e-3
                arena.grad[half:].copy_(want[half:])
            for i in range(len(red.buckets)):
                red._launch(i)
            red._wait_all()                                   # the current stream waits; reading on it is ordered behind the collectives
            got = arena.grad.clone()
            torch.cuda.synchronize()
            res[mode] = bool(torch.equal(got, want))
            # twice in a row (events re-recorded, pending flags reset)
            arena.grad.mul_(2)
            for i in range(len(red.buckets)):
                red._launch(i)
            red._wait_all()
            torch.cuda.synchronize()
            res[mode + "_again"] = bool(torch.equal(arena.grad, want * 2))
            red.close()
            res[mode + "_closed"] = red.native is None
        torch.save(res, os.path.join(outdir, "native.pt"))
    finally:
        dist.destroy_process_group()


def test_native_reducer_executes_over_rccl(dev, tmp_path):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    res = torch.load(os.path.join(str(tmp_path), "native.pt"))
    assert all(res.values()) and len(res) == 8, res


def _engine_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, ROOT)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    try:
        import random
        from types import SimpleNamespace
        from valor_amd import synth
        from valor_amd.engine import TrainEngine
        from valor_amd.model.valor import VALOR
        task = "pt_contra%tva%tv%ta_caption%tva%tv%ta_mlm%tva"
        spec = synth.tiny_spec()
        sd = synth.make_state_dict(spec, seed=3, w_std=0.05)
        batch = synth.make_batch(spec, batch=2, frames=2, audio_slices=1, txt_len=32, seed=4)
        flats, natives = [], []
        for native in ("1", "0"):
            os.environ["VALOR_REDUCER_NATIVE"] = native
            m = VALOR({"dropout": 0.0, "drop_path_rate": 0.0}, spec=spec, dtype=torch.bfloat16, device="cuda:0")
            m.load_state_dict(sd, strict=True)
            opts = SimpleNamespace(learning_rate=1e-3, weight_decay=0.01, clip_lr=1e-3, clip_lr_text=1e-3, new_lr=0.0, decoder_lr=-1,
                                   betas=[0.9, 0.98], warmup_ratio=0.1, num_train_steps=10, scheduler="warmup_linear", grad_norm=5.0)
            eng = TrainEngine(m, opts, manage_gc=False)
            start = m.arena.flat.clone()
            natives.append(eng.reducer.native is not None)
            for step in range(3):                       # step 1 learns the used parameters (synchronous path), 2 and 3 launch from the hooks
                random.seed(10 + step)
                eng.train_step(batch, task)
            torch.cuda.synchronize()
            flats.append(m.arena.flat.clone())
            eng.reducer.close()
        torch.save({"natives": natives, "equal": bool(torch.equal(flats[0], flats[1])), "moved": not torch.equal(flats[0], start)},
                   os.path.join(outdir, "engine.pt"))
    finally:
        dist.destroy_process_group()


def test_train_steps_through_the_native_reducer_match_the_torch_distributed_path(dev, tmp_path):
    """TrainEngine with VALOR_REDUCER_NATIVE=1 inside a one-rank RCCL process group: the autograd hooks launch the buckets on the
    library's reducer (first step: all at once; later steps: as the gradients complete, two compute streams), the optimizer's stream
    waits stream-side; three optimizer steps end in bit-identical parameters to the same steps without it"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_engine_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    res = torch.load(os.path.join(str(tmp_path), "engine.pt"))
    assert res == {"natives": [True, False], "equal": True, "moved": True}, res


def test_reducer_entry_points_validate_their_arguments(dev):
    import ctypes
    from valor_amd import lib
    so = lib.load()
    h = ctypes.c_void_p()
    ident = (ctypes.c_char * 128)()
    offs, cnts = (ctypes.c_int64 * 1)(0), (ctypes.c_int64 * 1)(16)
    z = torch.zeros(16, device=dev, dtype=torch.bfloat16)
    assert so.valor_reducer_unique_id(None) == -1
    assert so.valor_reducer_create(ctypes.byref(h), ident, 0, 1, 0, None, offs, cnts, 1, 0) == -1          # no arena
    assert so.valor_reducer_create(ctypes.byref(h), ident, 2, 1, 0, z.data_ptr(), offs, cnts, 1, 0) == -1  # rank >= world
    assert so.valor_reducer_create(ctypes.byref(h), ident, 0, 1, 0, z.data_ptr(), offs, cnts, 1, 7) == -1  # unknown mode
    assert so.valor_reducer_launch_bucket(None, 0, None, 0) == -1 and so.valor_reducer_wait(None, None) == -1
    assert so.valor_reducer_destroy(None) == 0
