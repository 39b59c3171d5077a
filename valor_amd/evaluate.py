"""Evaluation of a pretraining run (first slice): test.py::validate_pt -- masked-token accuracy of the caption /
mlm passes (test.py:462-533, the "argmax token ids" of the north star) and the in-batch retrieval recall from the fine-grained
score matrices of the whole validation set (test.py:534-660, 714-774) -- on the same HIP kernels as training: the model's
compute_loss=False branch produces scores / features, the all-pairs similarity is the bf16 / fp32 GEMM and the per-pair reduction
is valor_fine_scores (rectangular, forward only). The reference's slicing of large sets (pretrain.py:178-186: rows of A in slices of
100 once B has more than 1200 items) is kept, with the slice size a parameter (288 GB of HBM hold far larger tiles)."""
import math

import torch

from . import kernels as K
from . import lib, ops
from .lib import ACT_RELU


def _fine_weights(model, name, feat):
    """fine_weight_mapper[name](feat).squeeze(2) (pretrain.py:104-116), no grad"""
    P = model.P
    h = ops.linear(feat, P[f"{name}_fine_weight.0.weight"], P[f"{name}_fine_weight.0.bias"], ACT_RELU)
    return ops.rowdot(h, P[f"{name}_fine_weight.2.weight"], P[f"{name}_fine_weight.2.bias"]).float().squeeze(-1).contiguous()


def _fine_matrix_slice(featA, featB, maskA, maskB, wA_raw, wB_raw):
    NA, T, D = featA.shape
    NB, Nv = featB.shape[:2]
    dev = featA.device
    f32 = dict(dtype=torch.float32, device=dev)
    fa, fb = featA.contiguous().view(NA * T, D), featB.contiguous().view(NB * Nv, D)
    ldS = (NB * Nv + 7) // 8 * 8
    S = torch.empty((NA * T, ldS), **f32)
    K.gemm(fa, fb, out=S[:, :NB * Nv], out_dtype=torch.float32, splitk=False)
    wA, wB = torch.empty((NA, T), **f32), torch.empty((NB, Nv), **f32)
    st = K._stream()
    lib.call("valor_fine_weight_softmax", st, wA_raw.data_ptr(), maskA.data_ptr(), wA.data_ptr(), NA, T)
    lib.call("valor_fine_weight_softmax", st, wB_raw.data_ptr(), maskB.data_ptr(), wB.data_ptr(), NB, Nv)
    score = torch.empty((NA, NB), **f32)
    lib.call("valor_fine_scores", st, S.data_ptr(), ldS, maskA.data_ptr(), maskB.data_ptr(), wA.data_ptr(), wB.data_ptr(), score.data_ptr(),
             NA, NB, T, Nv)
    return score


@torch.no_grad()
def compute_fine_matrix(featA, featB, maskA, maskB, weightA, weightB, slice_rows=100, slice_above=1200):
    """VALOR.compute_fine_matrix (pretrain.py:178-189): [NA, NB] scores; rows of A in slices once B exceeds `slice_above` items."""
    maskA, maskB = maskA.float().contiguous(), maskB.float().contiguous()
    weightA, weightB = weightA.float().contiguous(), weightB.float().contiguous()
    if featB.shape[0] > slice_above:
        out = []
        for i in range(math.ceil(featA.shape[0] / slice_rows)):
            sl = slice(i * slice_rows, (i + 1) * slice_rows)
            out.append(_fine_matrix_slice(featA[sl], featB, maskA[sl], maskB, weightA[sl], weightB))
        return torch.cat(out, dim=0)
    return _fine_matrix_slice(featA, featB, maskA, maskB, weightA, weightB)


def compute_metric_ret(score_matrix, ids, ids_txt):
    """test.py:714-774 without dual softmax / text-retrieval direction: rank of the ground-truth video for every text."""
    order = score_matrix.sort(dim=-1, descending=True)[1].tolist()
    rank = torch.tensor([order[i].index(ids.index(ids_txt[i])) for i in range(len(ids_txt))], dtype=torch.float32)
    r1, r5, r10 = [(rank < k).sum().item() / len(ids_txt) for k in (1, 5, 10)]
    return {"forward_recall": f"{round(r1 * 100, 1)}/{round(r5 * 100, 1)}/{round(r10 * 100, 1)}", "forward_ravg": round((r1 + r5 + r10) / 3 * 100, 1),
            "forward_medianR": torch.median(rank).item() + 1, "forward_meanR": torch.mean(rank).item() + 1}


@torch.no_grad()
def validate_pt(model, loader, task):
    """test.py::validate_pt (:404-665): `loader` yields THIS RANK's batches of valor_collate (+ 'ids_txt'); returns the val_log dict
    (caption_acc_* / mlm_acc_* rounded to 2 digits, t2v / t2va / t2a forward recall strings).
    Under data parallelism (torch.distributed initialised, world > 1) every rank scores its shard and the reference's collectives run:
    ids / ids_txt and the hit / word counters through all_gather_list (test.py:275-276, 496-518), features and tokens through
    ddp_allgather (:279-290; per-rank sizes may differ). The recall is computed on the gathered set; the reference does that on rank 0
    only and leaves the other ranks' val_log without it -- here every rank returns the full log.
    Kept quirk: the mlm hit counters are selected by the CAPTION group list (test.py:484-492)."""
    from . import dist as vdist
    model.eval()
    mlm_task, caption_task, contra_task = [], [], []
    for i in task.split("_"):
        if "mlm" in i:
            mlm_task = i.split("%")[1:]
        elif "caption" in i:
            caption_task = i.split("%")[1:]
        elif "contra" in i:
            contra_task = i.split("%")[1:]
    n_word = {"caption": 0, "mlm": 0}
    hits = {}
    feats = {"feat_t": [], "feat_v": [], "feat_a": [], "txt_tokens": []}
    ids, ids_txt = [], []
    for batch in loader:
        ev = model(batch, task=task, compute_loss=False)
        if contra_task:
            for k in feats:
                feats[k].append(ev[k])
            ids += list(batch["ids"])
            ids_txt += list(batch.get("ids_txt", batch["ids"]))
        for tag, groups in (("caption", caption_task), ("mlm", mlm_task)):
            if not groups:
                continue
            lab = ev[f"txt_labels_{tag}"]
            lab = lab[lab != -1].to(model.device)
            n_word[tag] += lab.numel()
            for g in ("tva", "tv", "ta"):
                if g in caption_task and f"{tag}_scores_{g}" in ev:
                    hits[f"{tag}_{g}"] = hits.get(f"{tag}_{g}", 0) + int((ev[f"{tag}_scores_{g}"].max(dim=-1)[1] == lab).sum().item())
    if vdist.is_dist():
        # test.py:496-518: sum(all_gather_list(counter)) per counter -- one object collective for all of them here
        every = vdist.all_gather_list((n_word, hits))
        n_word = {k: sum(nw[k] for nw, _ in every) for k in n_word}
        hits = {k: sum(h.get(k, 0) for _, h in every) for k in sorted({k for _, h in every for k in h})}
        if contra_task:
            ids = [j for part in vdist.all_gather_list(ids) for j in part]               # test.py:275-276
            ids_txt = [j for part in vdist.all_gather_list(ids_txt) for j in part]
            for k in feats:                                                              # test.py:279-290
                if feats[k] and feats[k][0] is not None:
                    feats[k] = [vdist.ddp_allgather(torch.cat([t.to(model.device) for t in feats[k]], dim=0))]
    val_log = {}
    for tag, groups in (("caption", caption_task), ("mlm", mlm_task)):
        for g in ("tva", "tv", "ta"):
            if g in groups and f"{tag}_{g}" in hits:
                val_log[f"{tag}_acc_{g}"] = round(hits[f"{tag}_{g}"] / n_word[tag], 2)
    if contra_task and model.spec.contra_type == "coarse":            # test.py:640-660: plain similarity matrices of the pooled features
        from . import kernels as K, ops
        cat = lambda k: torch.cat(feats[k], dim=0).contiguous() if feats[k] and feats[k][0] is not None else None
        ft, fv, fa = cat("feat_t"), cat("feat_v"), cat("feat_a")
        sim = lambda a, b: K.gemm(a, b, out_dtype=torch.float32)
        with torch.no_grad():
            if "tv" in contra_task:
                val_log["t2v_recall"] = compute_metric_ret(sim(ft, fv).cpu(), ids, ids_txt)["forward_recall"]
            if "tva" in contra_task:
                if model.spec.late_fusion:
                    sm = sim(ft, fv) + sim(ft, fa)
                else:
                    fva = ops.l2_normalize(ops.linear(torch.cat((fv, fa), dim=-1), model.P["va_fusion.weight"], model.P["va_fusion.bias"]))
                    sm = sim(ft, fva)
                val_log["t2va_recall"] = compute_metric_ret(sm.cpu(), ids, ids_txt)["forward_recall"]
            if "ta" in contra_task:
                val_log["t2a_recall"] = compute_metric_ret(sim(ft, fa).cpu(), ids, ids_txt)["forward_recall"]
        return val_log
    if contra_task:
        cat = lambda k: torch.cat(feats[k], dim=0) if feats[k] and feats[k][0] is not None else None
        ft, fv, fa = cat("feat_t"), cat("feat_v"), cat("feat_a")
        tok = torch.cat([t.to(model.device) for t in feats["txt_tokens"]], dim=0)
        maskA = (tok != 0).float()
        wt = _fine_weights(model, "text", ft)
        ones = lambda f: torch.ones(f.shape[:2], dtype=torch.float32, device=model.device)
        if "tv" in contra_task:
            val_log["t2v_recall"] = compute_metric_ret(compute_fine_matrix(ft, fv, maskA, ones(fv), wt, _fine_weights(model, "video", fv)).cpu(), ids, ids_txt)["forward_recall"]
        if "tva" in contra_task and model.spec.late_fusion:          # test.py:571-579: unit token weights, the tv and ta matrices summed
            sm = compute_fine_matrix(ft, fv, maskA, ones(fv), ones(ft), ones(fv)) + compute_fine_matrix(ft, fa, maskA, ones(fa), ones(ft), ones(fa))
            val_log["t2va_recall"] = compute_metric_ret(sm.cpu(), ids, ids_txt)["forward_recall"]
        elif "tva" in contra_task:
            fva = torch.cat((fv, fa), dim=1)
            wva = torch.cat((_fine_weights(model, "video", fv), _fine_weights(model, "audio", fa)), dim=1)
            val_log["t2va_recall"] = compute_metric_ret(compute_fine_matrix(ft, fva, maskA, ones(fva), wt, wva).cpu(), ids, ids_txt)["forward_recall"]
        if "ta" in contra_task:
            val_log["t2a_recall"] = compute_metric_ret(compute_fine_matrix(ft, fa, maskA, ones(fa), wt, _fine_weights(model, "audio", fa)).cpu(), ids, ids_txt)["forward_recall"]
    return val_log
