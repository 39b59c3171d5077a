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
