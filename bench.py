"""bench.py -- VALOR-base tri-modal pretraining step on N MI355X of one node (data parallel, weak scaling).

  python bench.py --gpus N --steps K --warmup W          (N > 1: one rank per GPU -- under torch.distributed.run, or, started
                                                          bare, bench.py re-executes itself under it with N ranks)

Workload (BASELINE.json configs[1]/[2]): CLIP-ViT-B/16 + CLIP-text + AST + BERT-base decoder, bf16 compute
with fp32 master weights, task pt_contra%tva%tv%ta_caption%tva%tv%ta_mlm%tva (MGA + MGC + MLM), per-GPU batch
64, 8 frames x 224^2, 2 x 5.12 s audio slices, 32 text tokens, dropout 0.1 (the reference's training setting),
synthetic inputs resident in HBM, random-init weights. A step = forward + backward + gradient all-reduce +
global-norm clip + fused AdamW. Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import json
import os
import random
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TASK = "pt_contra%tva%tv%ta_caption%tva%tv%ta_mlm%tva"
PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PMC_FILE = next((f for f in ("r06_pmc_gemm_traffic.json", "r05_pmc_gemm_traffic.json", "r04_pmc_gemm_traffic.json", "r03_pmc_gemm_traffic.json", "r02_pmc_gemm_traffic.json", "r01_pmc_gemm_traffic.json")
                 if os.path.exists(os.path.join(ROOT, "profiles", f))), "r01_pmc_gemm_traffic.json")


def necessary_flops_per_sample(spec, frames, audio_slices, txt_len, n_cap_groups=3, mlm_prompt=10, mask_cap=0.6, mask_mlm=0.15):
    """fwd+bwd matmul FLOPs per sample with shared cross-K/V (SURVEY.md 8d formula); bwd = 2 x fwd."""
    H = spec.hidden
    La = spec.aud_tokens
    if spec.video_encoder == "swin":
        # VideoSwin (videoswin.py): patch embed K = 3*2*4*4, per block 12 C^2 MACs per token + window attention over
        # N = min(F, wd) * wh * ww slots, PatchMerging 4C -> 2C, then Linear(C_out -> hidden) of modeling.py:348
        side = spec.resolution // 4
        C = spec.swin_embed
        vit = 2 * frames * side * side * 96 * C
        N = min(frames, spec.swin_window[0]) * spec.swin_window[1] * spec.swin_window[2]
        for li, depth in enumerate(spec.swin_depths):
            tokens = frames * side * side
            vit += depth * (2 * tokens * 12 * C * C + 4 * tokens * min(N, tokens) * C)
            if li + 1 < len(spec.swin_depths):
                vit += 2 * (tokens // 4) * 4 * C * 2 * C
                C, side = 2 * C, side // 2
        Lv = side * side
        vit += 2 * frames * Lv * C * H
    else:
        W = spec.vis_width
        Lv = spec.vis_tokens
        p_vit = spec.vis_layers * 12 * W * W + 3 * spec.patch ** 2 * W
        vit = frames * (2 * Lv * p_vit + spec.vis_layers * 4 * Lv * Lv * W)
        if W != H:
            vit += 2 * frames * Lv * W * H                                                         # hidden_trans_video_multimodal
    if spec.txt_encoder == "bert":
        txt = spec.layers * (2 * txt_len * (4 * H * H + 2 * H * spec.inter) + 4 * txt_len ** 2 * H)   # BERT text pass, no cross-attention
    else:
        p_txt = spec.txt_layers * 12 * spec.txt_width ** 2
        txt = 2 * txt_len * p_txt + spec.txt_layers * 4 * txt_len ** 2 * spec.txt_width
    p_ast = spec.aud_layers * (4 * spec.aud_width ** 2 + 2 * spec.aud_width * spec.aud_inter) + spec.aud_patch ** 2 * spec.aud_width
    ast = audio_slices * (2 * La * p_ast + spec.aud_layers * 4 * La * La * spec.aud_width)
    Sv, Sa = frames * Lv, audio_slices * La

    def dec(T, Skv):
        return spec.layers * (2 * T * (4 * H * H + 2 * H * H + 2 * H * spec.inter) + 4 * T * T * H + 4 * T * Skv * H)
    decoder = dec(txt_len, Sv + Sa) + dec(txt_len, Sv) + dec(txt_len, Sa) + dec(txt_len + mlm_prompt, Sv + Sa)
    cross_kv = spec.layers * 2 * (Sv + Sa) * 2 * H * H
    n_mask = (n_cap_groups * mask_cap + mask_mlm) * (txt_len * 0.55)
    head = n_mask * 2 * (H * H + H * spec.vocab)
    fwd = vit + ast + txt + decoder + cross_kv + head
    return 3.0 * fwd


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sample_batch=2, frames=8, audio_slices=2, variant="clip", timed_steps=3):
    """The reference's CPU path on the host cores (BASELINE.md section 3 protocol: all cores, fp32, 1 warm-up + 3 timed full steps of
    forward + backward + clip + AdamW on a bounded sample -- batch 2 of the benchmarked geometry). Where /root/reference exists (the
    build container) the UNMODIFIED reference modules run through oracle/ref_harness.py (kind "reference"); on the GPU box, where the
    reference tree does not travel, the oracle's restatement of the same step runs (kind "port")."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from valor_amd import synth
    spec = {"clip": synth.base_spec, "swin": synth.swin_spec, "large": synth.large_spec, "clip_large": synth.clip_large_spec}[variant]()
    sd = synth.make_state_dict(spec, seed=50)
    batch = synth.make_batch(spec, batch=sample_batch, frames=frames, audio_slices=audio_slices, txt_len=32, seed=50)
    kind = "port"
    step = None
    if variant in ("clip", "swin"):
        try:
            import ref_harness
            if ref_harness.available():
                ropts = None if variant == "clip" else ref_harness.default_opts(video_encoder_type="videoswin_base_k400_22k",
                                                                                 txt_encoder_type="bert_base_uncased")
                ref = ref_harness.build_reference(ropts, state_dict=sd, dropout=0.0)
                from easydict import EasyDict
                from optim.misc import build_optimizer
                opt = build_optimizer(ref, EasyDict(learning_rate=1e-4, weight_decay=0.01, clip_lr=5e-7, clip_lr_text=5e-7, new_lr=0.0,
                                                    decoder_lr=-1, new_params_name=[], optim="adamw", betas=[0.9, 0.98]))

                def step():
                    opt.zero_grad()
                    sum(ref(batch, task=TASK, compute_loss=True).values()).backward()
                    torch.nn.utils.clip_grad_norm_(ref.parameters(), 5.0)
                    opt.step()
                kind = "reference"
        except Exception:
            step = None
    if step is None:
        import valor_oracle as VO
        sd_o = VO.trainable_copy(sd)
        orc = VO.Oracle(spec, sd_o, dropout_p=0.0, vocab_tokens=synth.synthetic_vocab(spec.vocab))
        params = {k: v for k, v in sd_o.items() if v.requires_grad and not VO.is_alias_key(k)}
        groups = {k: VO.param_group_of(k) for k in params}
        lrs, wds = VO.group_hparams(1e-4, 0.01)
        state = {}

        def step():
            for p in params.values():
                p.grad = None
            out = orc.forward_pt(batch, TASK, compute_loss=True)
            sum(out.values()).backward()
            grads = {k: p.grad for k, p in params.items() if p.grad is not None}
            VO.clip_grad_norm(grads, 5.0)
            with torch.no_grad():
                VO.adamw_step(params, grads, state, lrs, wds, groups)
    random.seed(50)
    times = []
    for i in range(1 + timed_steps):                  # step 0 = warm-up (pages the code in, spins the intra-op pool up, allocates the Adam state)
        t0 = time.time()
        step()
        if i:
            times.append(time.time() - t0)
    mean = sum(times) / len(times)
    what = "the UNMODIFIED reference modules (oracle/ref_harness.py)" if kind == "reference" else \
        "oracle/valor_oracle.py (the reference's CPU path restated; /root/reference does not exist on this box)"
    return {"value": round(sample_batch / mean, 4), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": kind,
            "cpu_model": _cpu_model(), "timed_steps": timed_steps, "step_seconds": [round(t, 2) for t in times],
            "value_min": round(sample_batch / max(times), 4), "value_max": round(sample_batch / min(times), 4),
            "sample": f"{timed_steps} timed full steps (fwd+bwd+clip+AdamW) after 1 warm-up step of {what}, fp32, batch {sample_batch}, "
                      f"{frames} frames, {audio_slices} audio slices, 32 tokens; mean {mean:.1f} s per step"}


def sim_world(model, spec, world, per_gpu_batch, frames, audio_slices, steps=5):
    """What data parallelism ADDS to a rank's step besides the collectives, measured on this one GPU: every rank computes the
    contrastive block on the GATHERED features (global batch = world x per-GPU batch; utils/distributed.py:38-72, pretrain.py:278-370),
    i.e. world^2 the pairs of the single-GPU step. Times forward + backward of the three MGA groups (tva, tv, ta) on synthetic gathered
    features at the local and at the global batch."""
    from valor_amd import ops
    dev = model.device
    D, T = spec.cdim, 32
    res = {}
    for tag, B in (("local", per_gpu_batch), ("global", world * per_gpu_batch)):
        g = torch.Generator(device="cpu").manual_seed(7)
        mk = lambda n: torch.nn.functional.normalize(torch.randn(B, n, D, generator=g), dim=-1).to(dev, model.dtype).requires_grad_(True)
        ft, fv, fa = mk(T), mk(frames), mk(audio_slices)
        wt, wv, wa = (torch.randn(B, n, generator=g).to(dev).requires_grad_(True) for n in (T, frames, audio_slices))
        maskA = (torch.rand(B, T, generator=g) < 0.7).float().to(dev)
        maskA[:, 0] = 1
        ones = lambda f: torch.ones(f.shape[:2], dtype=torch.float32, device=dev)
        k = torch.tensor(14.3, device=dev, requires_grad=True)

        def block():
            fB, wB = torch.cat((fv, fa), dim=1), torch.cat((wv, wa), dim=1)
            ls = [ops.fine_contrastive(ft, fB, wt, wB, maskA, ones(fB), k), ops.fine_contrastive(ft, fv, wt, wv, maskA, ones(fv), k),
                  ops.fine_contrastive(ft, fa, wt, wa, maskA, ones(fa), k)]
            (sum(ls) / 3).backward()
        block()
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            block()
        e1.record()
        torch.cuda.synchronize()
        res[tag] = {"batch": B, "ms": round(e0.elapsed_time(e1) / steps, 3), "peak_extra_mb": round((torch.cuda.max_memory_allocated() - base) / 2 ** 20, 1)}
    return {"world": world, "contrastive_ms_at_local_batch": res["local"]["ms"], "contrastive_ms_at_global_batch": res["global"]["ms"],
            "detail": res, "what": "forward + backward of the tva / tv / ta fine-grained contrastive groups on synthetic gathered features "
                                   "(every rank runs this on the global batch)"}


def comm_probe(dev, world, backend, bucket_bytes=48 << 20, gather_bytes=3_400_000, reps=10):
    """First-contact numbers of a multi-GPU node, taken before the timed region: one gradient bucket (48 MiB, the reducer's size) under
    a plain all-reduce and under reduce-scatter + all-gather (the two modes of valor_amd/dist.Reducer), and the step's packed feature
    all-gather (3.4 MB). Per call: microseconds (max over ranks) and bus bandwidth = 2 (N-1)/N (all-reduce) or (N-1)/N (all-gather) x
    bytes / time. Returns the dict that goes into the bench line as `comm_probe`."""
    import torch.distributed as dist
    n = bucket_bytes // 2
    n -= n % (world * 1024)
    x = torch.zeros(n, dtype=torch.bfloat16, device=dev)
    shard = x.view(world, -1)[dist.get_rank()]
    g_in = torch.zeros(max(gather_bytes // 2 // world, 1), dtype=torch.bfloat16, device=dev)
    g_out = torch.zeros(g_in.numel() * world, dtype=torch.bfloat16, device=dev)

    def allreduce():
        dist.all_reduce(x)

    def rs_ag():
        dist.reduce_scatter_tensor(shard, x)
        dist.all_gather_into_tensor(x, shard)

    def gather():
        if backend == "nccl":
            dist.all_gather_into_tensor(g_out, g_in)
        else:
            dist.all_gather(list(g_out.chunk(world)), g_in)

    def time_it(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    out = {"bucket_mib": round(n * 2 / 2 ** 20, 1), "reps": reps}
    cases = [("allreduce", allreduce, 2.0 * (world - 1) / world * n * 2)]
    if backend == "nccl":                      # gloo has no reduce_scatter
        cases.append(("rs_ag", rs_ag, 2.0 * (world - 1) / world * n * 2))
    cases.append(("packed_allgather", gather, (world - 1) / world * g_out.numel() * 2))
    for name, fn, bus_bytes in cases:
        try:
            t = time_it(fn)
            out[name] = {"us": round(t * 1e6, 1), "bus_GBps": round(bus_bytes / t / 1e9, 1)}
        except Exception as e:                 # a probe must never end the bench
            out[name] = {"error": repr(e)}
    # what RCCL said about the communicator (NCCL_DEBUG=INFO into a per-rank file, set by main() before the process group exists)
    path = os.environ.get("NCCL_DEBUG_FILE", "").replace("%h", "").replace("%p", str(os.getpid()))
    try:
        keys = ("nChannels", "Connected all", "Trees", "Using network", "comm ", "NCCL_ALGO", "P2P")
        lines = [l.strip()[-160:] for l in open(path) if any(k in l for k in keys)]
        out["rccl_log"] = lines[:6] + (["..."] if len(lines) > 12 else []) + lines[-6:] if len(lines) > 12 else lines
    except Exception:
        out["rccl_log"] = None
    return out


class GemmTimer:
    """HIP-event timing of valor_gemm launches on their own stream, grouped by (kernel family, layout).
    Every launch bracketed by an event pair would serialise the whole pipeline (an event is a barrier packet; ~1.9k
    GEMM launches per step), inflating every duration. So only every `stride`-th launch is timed, with a rotating
    offset per instrumented step; a timed launch carries a few microseconds of event overhead, nothing else changes."""
    FAMILY = {0: "gemm_kernel(128x128 reg-staged)", 1: "gemm_glds_kernel(128x128 LDS-DMA)", 2: "gemm_glds_kernel(128x128 LDS-DMA x2)",
              3: "gemm_8ph_kernel(256x256 8-phase)", 4: "gemm_8ph2_kernel(256x128 8-phase, 2 workgroups per CU)",
              5: "gemm_skinny_kernel(few rows, weights streamed once)"}

    def __init__(self, stride=4):
        self.records = []
        self.enabled = False
        self.stride, self.count, self.phase = stride, 0, 0

    def install(self):
        from valor_amd import kernels as K, lib
        so = lib.load()
        orig = K.gemm
        timer = self

        def timed(a, b, *, trans_a=False, trans_b=False, **kw):
            if not timer.enabled:
                return orig(a, b, trans_a=trans_a, trans_b=trans_b, **kw)
            timer.count += 1
            if (timer.count + timer.phase) % timer.stride:
                return orig(a, b, trans_a=trans_a, trans_b=trans_b, **kw)
            M, Kd = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
            N = b.shape[1] if trans_b else b.shape[0]
            fam = so.valor_gemm_kernel_for(0 if a.dtype == torch.bfloat16 else 1, int(trans_a), int(trans_b), M, N, Kd,
                                           int(kw.get("dact_aux") is not None and not (kw.get("act", 0) & 16)))   # ACT_DERIV aux: a light epilogue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(a, b, trans_a=trans_a, trans_b=trans_b, **kw)
            e1.record()
            timer.records.append(((fam, ("T" if trans_a else "N") + ("T" if trans_b else "N")), 2.0 * M * N * Kd, e0, e1))
            return r
        K.gemm = timed

    def next_step(self):
        self.phase += 1
        self.count = 0

    def summary(self):
        agg = {}
        for var, fl, e0, e1 in self.records:
            d = agg.setdefault(var, [0.0, 0.0, 0])
            d[0] += fl; d[1] += e0.elapsed_time(e1) * 1e-3; d[2] += 1
        return {v: {"flops": f, "seconds": s, "launches": n, "TFLOPs": f / s / 1e12, "avg_us": s / n * 1e6} for v, (f, s, n) in agg.items()}


def time_generation(args, dev, max_len=30):
    """caption generation of the headline model (VALOR.generate_cap, model/pretrain.py:914-985) on args.batch clips, group
    'tva', greedy and beam-3 to `max_len` tokens (random weights never emit [SEP]: every row runs the full length), encoders included:
    captions/s with the self-attention K|V cache and the graphed decoding step (valor_amd/decode.py), one warm call + two timed."""
    import time
    from valor_amd import decode, synth
    from valor_amd.model.valor import VALOR
    spec = synth.base_spec()
    model = VALOR({"dropout": args.dropout}, spec=spec, dtype=torch.bfloat16, device=dev)
    model.load_state_dict(synth.make_state_dict(spec, seed=50), strict=True)
    batch = synth.make_batch(spec, batch=args.batch, frames=args.frames, audio_slices=args.audio_slices, txt_len=32, seed=50)
    batch["video_pixels"], batch["audio_spectrograms"] = batch["video_pixels"].to(dev), batch["audio_spectrograms"].to(dev)
    out = {"clips": args.batch, "max_generation_len": max_len, "group": "tva", "kv_cache": decode.kv_cache_enabled(), "unit": "captions/s"}
    with torch.no_grad():
        for name, beam in (("greedy", 1), ("beam3", 3)):
            decode.generate_cap(model, batch, ["tva"], beam_size=beam, max_generation_len=max_len)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                decode.generate_cap(model, batch, ["tva"], beam_size=beam, max_generation_len=max_len)
            torch.cuda.synchronize()
            out[name] = round(2 * args.batch / (time.perf_counter() - t0), 1)
    decode.release_sessions(model)
    return out


def time_variant(variant, args, dev, steps=3, warmup=2, frames=None, accum=1, batch=None, checkpointing=False):
    """`steps` timed OPTIMIZER steps (each `accum` micro-steps of args.batch samples, train_utils.py:311-317) of another model variant at
    the headline's batch and -- unless `frames` says otherwise -- clip length (one rank): samples/s, step time, step MFU on the necessary
    FLOPs of THAT variant, peak memory."""
    frames = frames or args.frames
    nb = batch or args.batch
    from types import SimpleNamespace
    from valor_amd import synth
    from valor_amd.engine import TrainEngine
    from valor_amd.model.valor import VALOR
    spec = {"swin": synth.swin_spec, "large": synth.large_spec, "clip_large": synth.clip_large_spec}[variant]()
    mopts = {"dropout": args.dropout, "checkpointing": bool(checkpointing)}
    if variant == "clip_large":
        mopts.update(use_task_prompt=True, contra_loss_ratio=1.5)
    torch.cuda.reset_peak_memory_stats()
    model = VALOR(mopts, spec=spec, dtype=torch.bfloat16, device=dev)
    sd = synth.make_state_dict(spec, seed=50)
    model.load_state_dict(sd, strict=True)
    opts = SimpleNamespace(learning_rate=1e-4, weight_decay=0.01, clip_lr=5e-7, clip_lr_text=5e-7, new_lr=0.0, decoder_lr=-1,
                           betas=[0.9, 0.98], warmup_ratio=0.1, num_train_steps=100000, scheduler="warmup_linear", grad_norm=5.0)
    engine = TrainEngine(model, opts, graphs=False)      # the variants keep eager issue (their numbers stay comparable across rounds; the 16-frame
    engine.optimizer.init_master_from(sd)                # large runs have no memory to spare for a second, private pool of encoder activations)
    del sd
    batch = synth.make_batch(spec, batch=nb, frames=frames, audio_slices=args.audio_slices, txt_len=32, seed=50)
    batch["video_pixels"] = batch["video_pixels"].to(dev)
    batch["audio_spectrograms"] = batch["audio_spectrograms"].to(dev)
    for _ in range(warmup * accum):
        engine.train_step(batch, TASK, accum_steps=accum)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(steps * accum):
        last = engine.train_step(batch, TASK, accum_steps=accum)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    sps = nb * accum * steps / el
    nf = necessary_flops_per_sample(spec, frames, args.audio_slices, 32)
    out = {"value": round(sps, 2), "unit": "samples/s", "ms_per_step": round(el / steps * 1e3, 2), "steps": steps, "warmup": warmup,
           "step_mfu": round(nf * sps / 1e12 / PEAK_BF16_TFLOPS, 4), "necessary_gflop_per_sample": round(nf / 1e9, 1),
           "per_gpu_batch": nb * accum, "micro_batch": nb, "accum_steps": accum, "checkpointing": bool(checkpointing), "frames": frames, "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
           "losses": {k: round(float(v.detach()) if torch.is_tensor(v) else float(v), 4) for k, v in last.items()}}
    del engine, model, batch
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node. Without a launcher (no WORLD_SIZE in the environment) "
                    "N > 1 re-executes this script under torch.distributed.run with N ranks")
    ap.add_argument("--sim-world", type=int, default=8, help="also time the contrastive block at the global batch of this many ranks "
                    "(single-GPU runs only; 0 = off)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5, help="untimed steps (the first ones allocate: static K|V buffers, the caching allocator's pools)")
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (config: 512 global / 8 GPUs)")
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--audio-slices", type=int, default=2)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--checkpointing", type=int, default=0, help="1 = the reference's `checkpointing` option (activation checkpointing of every video / "
                    "audio / CLIP-text encoder layer and decoder layer): BASELINE configs[4] as written is --variant large --frames 16 --batch 128 --checkpointing 1")
    ap.add_argument("--graphs", type=int, default=int(os.environ.get("VALOR_GRAPHS", "1")), help="1 (default) = the CLIP ViT / AST / CLIP text encoders replay "
                    "hipGraphs (valor_amd/graphs.py: forward + backward captured on their third step; dropout offsets from a device-resident counter; "
                    "bit-identical to eager issue, profiles/r06_graphs_ab_*.txt), 0 = eager issue")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--accum", type=int, default=1, help="micro-steps per optimizer step (gradient accumulation, train_utils.py:311-317): the per-GPU batch of "
                    "a step is --batch x --accum. BASELINE configs[4] (VALOR-large, 16 frames, global batch 1024 = 128 per GPU) runs as "
                    "--variant large --frames 16 --batch 64 --accum 2: 128 samples of 16 frames do not fit one forward-backward (182 GB at 64)")
    ap.add_argument("--no-variants", action="store_true", help="skip the short timed runs of --variant swin / large behind the headline region")
    ap.add_argument("--no-roofline", action="store_true", help="skip the instrumented steps behind the timed region (roofline = null)")
    ap.add_argument("--variant", choices=["clip", "swin", "large", "clip_large"], default="clip",
                    help="clip: config/pretrain-VALOR-base.json (BASELINE configs[1], the headline); swin: scripts/pretrain.sh "
                         "(VideoSwin-B + BERT text); large: BASELINE configs[3]/[4] (VideoSwin-L embed 192 / 2-2-18-2 + BERT-large "
                         "24 x 1024, the reference classes at large hyper-parameters; --frames 16 for configs[4]); clip_large: the "
                         "reference's shipped config/pretrain-VALOR-large.json (CLIP ViT-L/14 at 224 px + shared BERT-base, task prompt)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (args.gpus or 1) > 1:
        # no launcher: become one. One process per GPU, rendezvous on the loopback address (the container hostname may not resolve).
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus is None:
        args.gpus = world
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("VALOR_DIST_BACKEND", "nccl")      # "gloo" only to exercise the DP path where RCCL cannot run
        if backend == "nccl" and "NCCL_DEBUG" not in os.environ:
            # the communicator's own account of itself (channels, rings / trees, transports) for `comm_probe`: INIT lines only, to a file
            os.environ["NCCL_DEBUG"] = "INFO"
            os.environ["NCCL_DEBUG_SUBSYS"] = "INIT"
            os.environ["NCCL_DEBUG_FILE"] = f"/tmp/valor_rccl_{os.getpid()}.log"
        if backend == "nccl":
            if torch.cuda.device_count() < world:
                sys.exit(f"bench.py: {world} RCCL ranks need {world} GPUs, this node shows {torch.cuda.device_count()} (one process per GPU)")
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            torch.distributed.init_process_group(backend, rank=rank, world_size=world)
        assert torch.distributed.get_world_size() == args.gpus

    from valor_amd import synth
    from valor_amd.engine import TrainEngine
    from valor_amd.model.valor import VALOR
    from valor_amd.ops import DropoutState

    spec = {"clip": synth.base_spec, "swin": synth.swin_spec, "large": synth.large_spec, "clip_large": synth.clip_large_spec}[args.variant]()
    np.random.seed(50 + rank)        # VideoSwin stochastic-depth draws
    mopts = {"dropout": args.dropout, "checkpointing": bool(args.checkpointing)}
    if args.variant == "clip_large":              # config/pretrain-VALOR-large.json:14-15
        mopts.update(use_task_prompt=True, contra_loss_ratio=1.5)
    model = VALOR(mopts, spec=spec, dtype=torch.bfloat16, device=dev)
    sd = synth.make_state_dict(spec, seed=50)                   # same weights on every rank (DDP broadcast equivalent)
    model.load_state_dict(sd, strict=True)
    opts = SimpleNamespace(learning_rate=1e-4, weight_decay=0.01, clip_lr=5e-7, clip_lr_text=5e-7, new_lr=0.0, decoder_lr=-1,
                           betas=[0.9, 0.98], warmup_ratio=0.1, num_train_steps=100000, scheduler="warmup_linear", grad_norm=5.0)
    engine = TrainEngine(model, opts, graphs=bool(args.graphs))
    engine.optimizer.init_master_from(sd)
    del sd
    batch = synth.make_batch(spec, batch=args.batch, frames=args.frames, audio_slices=args.audio_slices, txt_len=32, seed=50 + rank)
    batch["video_pixels"] = batch["video_pixels"].to(dev)
    batch["audio_spectrograms"] = batch["audio_spectrograms"].to(dev)
    random.seed(50 + rank)
    DropoutState.reset(1234 + rank)

    timer = GemmTimer()
    timer.install()
    probe = None
    if world > 1:
        # first contact with a multi-GPU node: time one bucket under both reduce modes and the packed gather; unless VALOR_REDUCE pins the
        # mode, the timed region runs the faster one (every rank sees the same max-over-ranks times, so every rank decides alike)
        try:
            probe = comm_probe(dev, world, backend)
            ar, rs = probe.get("allreduce", {}).get("us"), probe.get("rs_ag", {}).get("us")
            if "VALOR_REDUCE" not in os.environ and engine.reducer.native is None and ar and rs and rs < 0.95 * ar:
                engine.reducer.mode = "rs_ag"
            probe["reduce_mode_chosen"] = engine.reducer.mode
        except Exception as e:
            probe = {"error": repr(e)}

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def one_step():
        out = None
        for _ in range(args.accum):
            out = engine.train_step(batch, TASK, accum_steps=args.accum)
        return out

    # Warm-up with a synchronisation in the middle: the timed region starts from an idle GPU, where the host (45 ms per step) runs two
    # steps ahead of the device (118 ms) before the launch queue throttles it -- the tensors that cross streams (the decoder's
    # [video | audio] input and its gradient, one MLP activation) are then alive in three steps at once and the caching allocator
    # grows by a segment each (round 3: 4 device allocations, 2 + 172 + 172 + 592 MiB, inside the timed region). Starting the last
    # warm-up steps from an idle GPU too puts that high-water mark into the warm-up.
    # the per-step tracing of the timed region (timing-enabled events, their first synchronisation) is on during the warm-up too: whatever
    # the runtime sets up on first use of those paths must not happen inside the first timed steps of a cold process
    engine.trace = []
    for i in range(args.warmup):
        if args.warmup >= 4 and i == args.warmup - 3:
            sync()
        one_step()
    sync()
    if engine.trace:
        engine.trace[0]["head"].elapsed_time(engine.trace[-1]["tail"])
    seg0 = torch.cuda.memory_stats().get("segment.all.allocated", 0)       # device allocations (hipMalloc) so far
    segs_before = {sg["address"] for sg in torch.cuda.memory_snapshot()}
    seg_trace = [] if os.environ.get("BENCH_SEG_TRACE") else None      # diagnostic: cumulative device allocations after every timed step
    engine.trace = []                    # per-step wait / launch times and head / tail events (valor_amd/engine.py)
    t0 = time.perf_counter()
    last = None
    step_ms = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        last = one_step()
        step_ms.append((time.perf_counter() - ts) * 1e3)                    # host time of the (asynchronous) step, no sync inside the region
        if seg_trace is not None:
            seg_trace.append(int(torch.cuda.memory_stats().get("segment.all.allocated", 0) - seg0))
    sync()
    elapsed = time.perf_counter() - t0
    trace, engine.trace = engine.trace, None
    # what the host did per optimizer step: launch_ms = time inside train_step minus the `max_ahead` throttle wait (pure issue time),
    # wait_ms = the throttle wait (the host was AHEAD of the device); and what the device saw: gpu_span_ms = first launch -> last
    # launch of a step on the step's stream, gpu_idle_ms = the gap between a step's last kernel and the next step's first one (> 0 only
    # if the host had not issued the next step in time: the device idled for the host)
    timing = {"launch_ms": [round(r["host_ms"] - r["wait_ms"], 1) for r in trace], "wait_ms": [round(r["wait_ms"], 1) for r in trace],
              "gpu_span_ms": [round(r["head"].elapsed_time(r["tail"]), 1) for r in trace],
              "gpu_idle_ms": [round(a["tail"].elapsed_time(b["head"]), 2) for a, b in zip(trace[:-1], trace[1:])]}
    if world > 1 and trace and trace[0].get("reduce"):
        ex = [r["reduce"][0].elapsed_time(r["reduce"][1]) for r in trace]
        timing["reduce_exposed_ms"] = [round(x, 2) for x in ex]
        timing["reduce_exposed_ms_per_step"] = round(sum(ex) / len(ex), 2)
    if timing["gpu_idle_ms"]:
        timing["gpu_idle_ms_per_step"] = round(sum(timing["gpu_idle_ms"]) / len(timing["gpu_idle_ms"]), 2)
        timing["launch_ms_per_step"] = round(sum(timing["launch_ms"]) / len(timing["launch_ms"]), 1)
        timing["wait_ms_per_step"] = round(sum(timing["wait_ms"]) / len(timing["wait_ms"]), 1)
    seg1 = torch.cuda.memory_stats().get("segment.all.allocated", 0)
    new_segs_mb = sorted(round(sg["total_size"] / 2 ** 20, 1) for sg in torch.cuda.memory_snapshot() if sg["address"] not in segs_before)
    # roofline pass: the SAME step, run right after the timed region, with a HIP-event pair (recorded on the launch
    # stream) around every 4th valor_gemm launch; 4 instrumented steps with a rotating offset cover every launch once.
    timer.enabled = True
    n_inst = 0 if args.no_roofline else 4
    # kernel durations are only meaningful when the kernel has the chip to itself: the instrumented steps run the encoders on ONE
    # stream (the timed region above runs them on two, valor_amd/streams.py)
    two_streams = os.environ.get("VALOR_ENCODER_STREAMS")
    os.environ["VALOR_ENCODER_STREAMS"] = "0"
    if n_inst and args.graphs:
        # ... and only when Python issues it: the instrumented steps run eagerly (a GEMM inside a replayed graph never passes the timer,
        # and an event pair recorded while a graph is being captured is not a timestamp)
        model.enable_graphs(False)
    timer.enabled = False
    for _ in range(2 if n_inst else 0):          # settle: the audio / text activations move from the side stream's allocator pool to this stream's (device
        one_step()                          # allocations stall the host, a starved GPU makes event pairs measure launch latency)
    sync()
    timer.enabled = True
    t1 = time.perf_counter()
    for _ in range(n_inst):
        one_step()
        timer.next_step()
    sync()
    inst_elapsed = (time.perf_counter() - t1) / max(n_inst, 1)
    timer.enabled = False
    if two_streams is None:
        del os.environ["VALOR_ENCODER_STREAMS"]
    else:
        os.environ["VALOR_ENCODER_STREAMS"] = two_streams
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    replicas_identical = None
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        # data parallelism keeps the replicas bit-identical (same reduced gradients, same optimizer): an exact integer checksum of the
        # parameter arena from every rank must agree
        cs = model.arena.flat.view(torch.int16).to(torch.int64).sum().reshape(1)
        allcs = [torch.zeros_like(cs) for _ in range(world)]
        torch.distributed.all_gather(allcs, cs)
        replicas_identical = all(int(c.item()) == int(allcs[0].item()) for c in allcs)
    elapsed = float(t.item())

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        sps = world * args.batch * args.accum * args.steps / elapsed
        gs = timer.summary()
        dom = max(gs, key=lambda v: gs[v]["seconds"]) if gs else None
        nf = necessary_flops_per_sample(spec, args.frames, args.audio_slices, 32)
        roof = None
        if dom:
            layout = {"NN": "forward x.W^T", "NT": "dgrad dY.W", "TT": "wgrad dY^T.X", "TN": "A^T.B^T"}
            name = lambda v: f"{GemmTimer.FAMILY[v[0]]} {v[1]} ({layout[v[1]]})"
            tot_f, tot_s = sum(d["flops"] for d in gs.values()), sum(d["seconds"] for d in gs.values())
            traffic = None
            try:    # HBM-side traffic of this kernel from the committed PMC passes (rocprofv3 cannot run inside the bench)
                pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))["kernels"].get(name(dom))
                if pmc:
                    traffic = {"bytes_per_launch": pmc["traffic_bytes"], "algorithmic_bytes": pmc["algorithmic_bytes"], "shape": pmc["shape"],
                               "source": f"profiles/{PMC_FILE} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE)"}
            except Exception:
                traffic = None
            roof = {"bound": "mfma", "kernel": name(dom), "achieved": round(gs[dom]["TFLOPs"], 1), "peak": PEAK_BF16_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(gs[dom]["TFLOPs"] / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                    "avg_launch_us": round(gs[dom]["avg_us"], 1), "launches_per_step": gs[dom]["launches"],
                    "all_gemm_kernels": {name(v): {"TFLOPs": round(d["TFLOPs"], 1), "avg_us": round(d["avg_us"], 1), "launches_per_step": d["launches"],
                                                   "share_of_step_time": round(d["seconds"] / inst_elapsed, 3)} for v, d in sorted(gs.items(), key=lambda kv: -kv[1]["seconds"])},
                    "all_gemms": {"TFLOPs": round(tot_f / tot_s / 1e12, 1), "frac": round(tot_f / tot_s / 1e12 / PEAK_BF16_TFLOPS, 4),
                                  "share_of_step_time": round(tot_s / inst_elapsed, 3)},
                    "measured": "algorithmic 2MNK FLOP / HIP-event duration, one event pair (on the launch stream) around every 4th valor_gemm "
                                "launch in %d instrumented steps with rotating offset run right after the timed region (each launch of a step "
                                "timed once; instrumented step %.1f ms)" % (n_inst, inst_elapsed * 1e3),
                    "step_mfu": round(nf * sps / world / 1e12 / PEAK_BF16_TFLOPS, 4),
                    "necessary_gflop_per_sample": round(nf / 1e9, 1)}
        arch = {"clip": "CLIP-B/16 + AST + BERT-base", "swin": "VideoSwin-B + BERT text + AST + BERT-base",
                "large": "VideoSwin-L + BERT-large text + AST + BERT-large", "clip_large": "CLIP-L/14@224 + BERT-base text + AST + BERT-base"}[args.variant]
        size = "VALOR-base" if args.variant in ("clip", "swin") else "VALOR-large"
        res = {"metric": f"pretrain samples/sec (V+A+T {args.variant})", "value": round(sps, 2), "unit": "samples/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 2), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "ranks": torch.distributed.get_world_size() if world > 1 else 1, "backend": backend,
               "replicas_identical": replicas_identical, "reduce_mode": engine.reducer.mode, "graphs": bool(args.graphs), "checkpointing": bool(args.checkpointing),
               "rccl_version": ".".join(str(x) for x in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None,
               "config": {"workload": f"{size} tri-modal ({arch}) pretrain step, MGA+MGC+MLM, "
                                      f"{args.frames} frames x 224^2, {args.audio_slices} x 5.12 s audio, 32 tokens",
                          "per_gpu_batch": args.batch * args.accum, "micro_batch": args.batch, "accum_steps": args.accum,
                          "global_batch": world * args.batch * args.accum, "parallelism": f"dp{world}",
                          "dropout": args.dropout, "task": TASK},
               "losses": {k: round(float(v.detach()) if torch.is_tensor(v) else float(v), 4) for k, v in last.items()},
               "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
               "reserved_mem_gb": round(torch.cuda.memory_reserved() / 2 ** 30, 1),
               "timed_region": {"device_allocations": int(seg1 - seg0), "new_segments_mb": new_segs_mb, "host_ms_per_step": [round(x, 1) for x in step_ms], **timing,
                                **({"allocations_after_step": seg_trace} if seg_trace is not None else {})},
               "roofline": roof}
        if probe is not None:
            res["comm_probe"] = probe
        if world == 1 and args.sim_world > 1:
            try:
                res["dp_sim"] = sim_world(model, spec, args.sim_world, args.batch, args.frames, args.audio_slices)
            except Exception as e:
                res["dp_sim"] = {"error": repr(e)}
        if world == 1 and args.variant == "clip" and not args.no_variants:
            # the other configurations this repository claims, driver-visible: a few timed steps each of the VideoSwin-B variant
            # (scripts/pretrain.sh) and of BASELINE configs[3] (VideoSwin-L + BERT-large), same batch / clip length, after the headline region
            del engine, model, batch
            gc.collect()
            torch.cuda.empty_cache()
            res["variants"] = {}
            # large_f16_accum2 = BASELINE configs[4]'s per-GPU share (VALOR-large, 16 frames, 128 samples per GPU of the global 1024) as
            # its recipe: two accumulated micro-steps of 64, 2 timed optimizer steps
            # large_f16_b128_ckpt = the same configuration AS WRITTEN: 128 samples of 16 frames in ONE forward-backward with the
            # reference's `checkpointing` option (every video / audio encoder layer re-run in backward), 2 timed steps
            for v, kw in (("swin", {}), ("large", {}), ("large_f16_accum2", dict(variant="large", frames=16, accum=2, steps=2, warmup=1)),
                          ("large_f16_b128_ckpt", dict(variant="large", frames=16, batch=2 * args.batch, checkpointing=True, steps=2, warmup=1))):
                try:
                    res["variants"][v] = time_variant(kw.pop("variant", v), args, dev, **kw)
                except Exception as e:
                    res["variants"][v] = {"error": repr(e)}
                gc.collect()
                torch.cuda.empty_cache()
            try:                       # the adjacent consumer of the same kernels, driver-visible: captions/s
                res["generation"] = time_generation(args, dev)
            except Exception as e:
                res["generation"] = {"error": repr(e)}
            gc.collect()
            torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(frames=args.frames, audio_slices=args.audio_slices, variant=args.variant)
            except Exception as e:      # the baseline is a reported number only; never fail the bench on it
                res["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
