"""The step after the path: DistMult triple scoring and the [B, N] score functions (csrc/scoring.hip).

Part of ``mr_gnas_amd.functional`` (autograd Functions over the C ABI, include/mrgnas.h): every Function enqueues HIP kernels of
libmrgnas_hip.so on torch's current stream through ctypes; every call site states the algorithmic bytes / flops of the launch."""
import torch

from .._lib import call, f32c, ptr, require_hip, stream_of
from .gcs import span_gcs
from .compose_gather import compose
from .row_linear import linear


class ScorePlan:
    """Index structures of a scoring batch of (s, r, o) triples: int32 indices for the forward and
    three span plans (by subject, by object, by relation) whose metadata carries the element id in
    the scale slot, so the backward can take the upstream gradient as an external scale."""

    def __init__(self, triplets, n_ent, n_rel):
        from ..graph import span_plan
        t = triplets.long()
        s, r, o = t[:, 0].contiguous(), t[:, 1].contiguous(), t[:, 2].contiguous()
        self.T, self.n_ent, self.n_rel = int(t.shape[0]), int(n_ent), int(n_rel)
        self.s32, self.r32, self.o32 = (x.to(torch.int32).contiguous() for x in (s, r, o))

        def packed(plan, xi, yi):
            from ..graph import span_meta
            return span_meta(plan, xi, yi, None, w_is_index=True)
        self.by_s, self.by_o, self.by_r = span_plan(s, n_ent), span_plan(o, n_ent), span_plan(r, n_rel)
        self.m_s = packed(self.by_s, o, r)        # g_ent[s] += g_t * ent[o] * rel[r]
        self.m_o = packed(self.by_o, s, r)        # g_ent[o] += g_t * ent[s] * rel[r]
        self.m_r = packed(self.by_r, s, o)        # g_rel[r] += g_t * ent[s] * ent[o]     (Y = ent as well)


class _DistMult(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ent, rel, sp):
        ent, rel = f32c(ent), f32c(rel)
        require_hip(ent, rel)
        D = ent.shape[1]
        score = torch.empty(sp.T, dtype=torch.float32, device=ent.device)
        call("mrg_distmult_score", (ptr(ent), ptr(rel), ptr(sp.s32), ptr(sp.r32), ptr(sp.o32), ptr(score), sp.T, D, stream_of(ent)),
             nbytes=sp.T * (12 * D + 16))
        ctx.sp = sp
        ctx.save_for_backward(ent, rel)
        return score

    @staticmethod
    def backward(ctx, g):
        ent, rel = ctx.saved_tensors
        sp = ctx.sp
        g = f32c(g)
        g_ent = span_gcs("mul", ent, rel, sp.m_s, sp.by_s, ext_scal=g)
        g_ent += span_gcs("mul", ent, rel, sp.m_o, sp.by_o, ext_scal=g)
        g_rel = span_gcs("mul", ent, ent, sp.m_r, sp.by_r, ext_scal=g)
        return g_ent, g_rel, None


def distmult_score(ent, rel, sp):
    """sum_c ent[s] * rel[r] * ent[o] per triple (reference models/model_search_lp.py:169-176)."""
    return _DistMult.apply(ent, rel, sp)


def distmult_scores_all(all_ent, sub_emb, rel_emb):
    """sf_DisMult_op (reference models/operations_lp.py:115-127): sigmoid((sub * rel) all_ent^T) as the compose kernel +
    the MFMA row GEMM with a sigmoid epilogue (all_ent [N, D] is the GEMM's weight operand: no transpose, no [B, N]
    pre-activation tensor)."""
    return linear(compose("mult", sub_emb, rel_emb), all_ent, None, "sigmoid")


class _TransE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, all_ent, sub, rel, gamma):
        all_ent, sub, rel = f32c(all_ent), f32c(sub), f32c(rel)
        require_hip(all_ent, sub, rel)
        B, D = sub.shape
        N = all_ent.shape[0]
        score = torch.empty(B, N, dtype=torch.float32, device=sub.device)
        call("mrg_transe_score_fwd", (ptr(all_ent), ptr(sub), ptr(rel), float(gamma), ptr(score), B, N, D, stream_of(sub)),
             nbytes=4 * (B * N + (N + 2 * B) * D))
        ctx.save_for_backward(all_ent, sub, rel, score)
        return score

    @staticmethod
    def backward(ctx, g):
        all_ent, sub, rel, score = ctx.saved_tensors
        g = f32c(g)
        B, D = sub.shape
        N = all_ent.shape[0]
        gent = torch.empty_like(all_ent) if ctx.needs_input_grad[0] else None
        gobj = torch.empty_like(sub) if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) else None
        call("mrg_transe_score_bwd", (ptr(all_ent), ptr(sub), ptr(rel), ptr(g), ptr(score), ptr(gent), ptr(gobj), B, N, D, stream_of(sub)))
        return gent, gobj, gobj, None


def transe_scores_all(all_ent, sub_emb, rel_emb, gamma):
    """sf_TransE_op (reference models/operations_lp.py:101-112): sigmoid(gamma - ||sub + rel - ent||_1) for all entities."""
    return _TransE.apply(all_ent, sub_emb, rel_emb, gamma)
