"""Dense linear on rows: the split-bf16 / exact-f32 MFMA row GEMM and its two gradients (csrc/linear.hip).

Part of ``mr_gnas_amd.functional`` (autograd Functions over the C ABI, include/mrgnas.h): every Function enqueues HIP kernels of
libmrgnas_hip.so on torch's current stream through ctypes; every call site states the algorithmic bytes / flops of the launch."""
import torch

from .. import _lib
from .._lib import call, f32c, ptr, require_hip, stream_of
from . import switches as SW
from ._base import ACT, _ws, _ws_bytes


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b, act):
        x, W, b = f32c(x), f32c(W), f32c(b)
        require_hip(x, W, b)
        rows, K = x.shape
        Nout = W.shape[0]
        if W.shape[1] != K:
            raise _lib.MrgnasError(f"linear: weight {tuple(W.shape)} does not match input width {K}")
        y = torch.empty(rows, Nout, dtype=torch.float32, device=x.device)
        gws = _ws(_ws_bytes("mrg_gemm_workspace_bytes", K, Nout), x)
        call("mrg_linear_fwd", (ptr(x), ptr(W), ptr(b), ptr(y), ptr(gws), rows, K, Nout, act, stream_of(x)),
             nbytes=4 * rows * (K + Nout) + 4 * K * Nout, flops=2 * rows * K * Nout)
        ctx.act, ctx.has_b = act, b is not None
        ctx.save_for_backward(x, W, y if act != 0 else None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, W, y = ctx.saved_tensors
        g = f32c(g)
        rows, K = x.shape
        Nout = W.shape[0]
        st = stream_of(x)
        gx = gW = gb = None
        work = dict(nbytes=4 * rows * (K + Nout) + 4 * K * Nout, flops=2 * rows * K * Nout)
        # a wide, short product ([B, N] scores against the whole entity table: Nout = N >> rows): both gradients reduce over
        # or stream along the N entity rows, so both run on the transposed score gradient g^T [N, B]
        wide = Nout > 1024
        need_w = ctx.needs_input_grad[1] or (ctx.has_b and ctx.needs_input_grad[2])
        wide_in = wide and SW.WIDE_BWD_INPUT and ctx.needs_input_grad[0] and rows <= 1024 and (rows <= 128 or (rows % 4 == 0 and K % 4 == 0))
        gT = None
        if wide and SW.WIDE_BWD_INPUT and (wide_in or not ctx.needs_input_grad[0]) and (wide_in or need_w):
            # the activation's derivative and the transposition in one pass over g and y (mrg_act_grad_transpose); g itself is not needed
            require_hip(g)
            gT = torch.empty(Nout, rows, dtype=torch.float32, device=x.device)
            call("mrg_act_grad_transpose", (ptr(g), ptr(y), ptr(gT), rows, Nout, ctx.act, st), nbytes=4 * rows * Nout * (3 if ctx.act else 2))
            g = None
        else:
            if ctx.act == 1:
                g = g * (y > 0)           # ReLU mask (elementwise; folded into the fused kernel later)
            elif ctx.act == 2:
                g = g * y * (1 - y)       # sigmoid
            if wide_in or (wide and need_w):
                gT = g.t().contiguous()
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            if wide_in:
                # gx = g W = (g^T)^T W is a reduction over N into a [B, K] block: the shape of a weight gradient (rows := N,
                # gY := g^T, X := W).  The row GEMM would give the whole N-long reduction to ceil(B / 128) workgroups
                # (36 ms at B = 256, N = 1 M, K = 256); the split-over-rows kernel spreads it over the chip.
                ws = _ws(_ws_bytes("mrg_linear_bwd_weight_workspace_bytes", Nout, K, rows), x)
                call("mrg_linear_bwd_weight", (ptr(gT), ptr(W), None, ptr(gx), None, ptr(ws), Nout, K, 0, rows, st), **work)
            else:
                wt = _ws(_ws_bytes("mrg_linear_bwd_input_workspace_bytes", K, Nout), x)
                call("mrg_linear_bwd_input", (ptr(g), ptr(W), ptr(gx), ptr(wt), rows, K, Nout, K, 0, st), **work)
        if need_w:
            gW = torch.empty_like(W)
            gb = torch.empty(Nout, dtype=torch.float32, device=x.device) if ctx.has_b else None
            if wide:
                # the split-over-rows weight-gradient kernel keeps all of gW's row tiles in registers and does not cover this
                # shape; here gW = g^T x is itself a tall-skinny row GEMM over the transposed operands
                xT = x.t().contiguous()
                gws = _ws(_ws_bytes("mrg_gemm_workspace_bytes", rows, K), x)
                call("mrg_linear_fwd", (ptr(gT), ptr(xT), None, ptr(gW), ptr(gws), Nout, rows, K, 0, st), **work)
                if gb is not None:
                    gb = gT.sum(1)
            else:
                ws = _ws(_ws_bytes("mrg_linear_bwd_weight_workspace_bytes", rows, K, Nout), x)
                call("mrg_linear_bwd_weight", (ptr(g), ptr(x), None, ptr(gW), ptr(gb), ptr(ws), rows, K, 0, Nout, st), **work)
        return gx, gW, gb, None


def linear(x, W, b=None, act=None):
    """act(x W^T + b) with exact-f32 MFMA (nn.Linear semantics)."""
    return _Linear.apply(x, W, b, ACT[act])


def module_linear(mod, x):
    """An nn.Linear module applied to HIP rows on the library's row GEMM (forward, input and weight gradient) instead of the
    vendor GEMM torch would pick: the entity projection and the cells' concat Linear (reference models/model_search_lp.py:131,
    models/cell_lp.py:186-188) are [N, .] x [., D] products whose Tensile kernels cost 60-120 us each at N = 14 541."""
    # The library path bypasses nn.Module.__call__: a module that carries forward (pre-)hooks, or a call under autocast, goes
    # through torch so that the hooks fire and the cast happens (advisor r4)
    hooked = bool(mod._forward_hooks or mod._forward_pre_hooks or getattr(mod, "_forward_hooks_with_kwargs", None)
                  or getattr(mod, "_forward_pre_hooks_with_kwargs", None))
    if (SW.NODE_LINEAR and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and not hooked
            and not torch.is_autocast_enabled()):
        return linear(x, mod.weight, mod.bias)
    return mod(x)
