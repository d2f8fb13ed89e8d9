"""GPU parity of the HIP operators (through the C ABI) against (1) the golden
vectors produced by the reference and (2) the CPU oracle on seeded inputs.
Tolerance: 1e-4 relative (BASELINE.json north_star), bit-exact for gathers."""
import numpy as np
import pytest
import torch

from conftest import load_golden, ops_inputs, sub
import mr_gnas_amd
from mr_gnas_amd import functional as K, graph as G, operations_lp as O
from oracle import ops as OO
from oracle.graph import OGraph

pytestmark = pytest.mark.gpu
DEV = "cuda"
OPS_CASES = ["ops_tiny_train", "ops_small_search", "ops_mid_train", "ops_d100_search", "ops_odd_train", "ops_r300_d64_search"]


def close(a, b, what, rtol=1e-4, atol=2e-5, rms_rtol=1e-4):
    """Two bounds: the largest error against the tensor's largest entry -- floored at 1.0, so for a tensor whose
    entries are << 1 this bound alone is an ABSOLUTE 1.2e-4 -- and the rms error against the tensor's rms, which scales with the
    tensor whatever its magnitude (a tensor of 1e-3-sized entries must agree to 1e-7, not to 1e-4)."""
    a = a.detach().cpu()
    scale = float(b.abs().max()) if b.numel() else 1.0
    err = float((a - b).abs().max()) if b.numel() else 0.0
    assert err <= atol + rtol * max(scale, 1.0), f"{what}: max err {err:.3e} (scale {scale:.3e})"
    if b.numel() and rms_rtol is not None:
        rms_b = float(b.double().square().mean().sqrt())
        rms_e = float((a.double() - b.double()).square().mean().sqrt())
        assert rms_e <= rms_rtol * rms_b + 1e-9, f"{what}: rms err {rms_e:.3e} against rms {rms_b:.3e} (ratio {rms_e / max(rms_b, 1e-300):.2e} > {rms_rtol:g})"


def dev_graph(z):
    return G.RelGraph(z["N"], z["src"], z["dst"], z["etype"], z["norm"], device=DEV)


def run_module(op, g, a, b, gout):
    a = a.to(DEV).requires_grad_(True)
    b = b.to(DEV).requires_grad_(True)
    out = op(g, a, b)
    out.backward(gout.to(DEV))
    z = torch.zeros_like
    return out, (a.grad if a.grad is not None else z(a)), (b.grad if b.grad is not None else z(b))


@pytest.mark.parametrize("case", OPS_CASES)
def test_ops_against_reference_golden(case):
    z = load_golden(case)
    g = dev_graph(z)
    tags = sorted({k.split("/")[0] for k in z if k.endswith("/out") and not k.startswith("sf_")})
    x, x_in, hr, xn, gM, gN = ops_inputs(z)
    for tag in tags:
        name = tag.split("@")[0]
        if tag.endswith("@node"):
            a, b, gout = xn, xn, gN
        elif name.startswith("a_"):
            a, b, gout = x, x_in, gN
        elif name.startswith("pre_"):
            a, b, gout = x, hr, gM
        else:
            a, b, gout = x, x_in, gM
        op = O.MIXED_OPS[name]({"feature_dim": z["D"], "drop_aggr": 0.0}).to(DEV)
        op.load_state_dict(sub(z, tag + "/param/"))          # reference state_dict loads unchanged
        out, ga, gb = run_module(op, g, a, b, gout)
        close(out, z[tag + "/out"], f"{case}:{tag} out")
        close(ga, z[tag + "/ga"], f"{case}:{tag} grad src_emb")
        close(gb, z[tag + "/gb"], f"{case}:{tag} grad src_emb_in")
        for k, p in op.named_parameters():
            got = p.grad if p.grad is not None else torch.zeros_like(p)
            close(got, z[f"{tag}/gparam/{k}"], f"{case}:{tag} grad {k}", rtol=2e-4, atol=5e-5)


def test_gather_is_bit_exact():
    gen = torch.Generator().manual_seed(0)
    for D in (8, 10, 64, 200, 256, 512):
        ent = torch.randn(301, D, generator=gen)
        rel = torch.randn(17, D, generator=gen)
        ei = torch.randint(0, 301, (1000,), generator=gen)
        ri = torch.randint(0, 17, (1000,), generator=gen)
        out = K.gather_rows(ent.to(DEV), ei.to(DEV).int())
        assert torch.equal(out.cpu(), ent[ei]), D
        for kind, f in (("sub", lambda a, b: a - b), ("mult", lambda a, b: a * b), ("add", lambda a, b: a + b)):
            out = K.gather_rows(ent.to(DEV), ei.to(DEV).int(), rel.to(DEV), ri.to(DEV).int(), kind)
            assert torch.equal(out.cpu(), f(ent[ei], rel[ri])), (D, kind)


def synth(N, T, R, seed, hub_frac=0.3):
    rng = np.random.default_rng(seed)
    p = 1.0 / np.arange(1, N + 1) ** 0.75
    p /= p.sum()
    s, o = rng.choice(N, size=T, p=p), rng.choice(N, size=T, p=p)
    o[: int(T * hub_frac)] = 1
    r = rng.integers(0, R, size=T)
    return np.stack([s, r, o], 1)


@pytest.mark.parametrize("N,T,R,D,order", [(500, 6000, 11, 200, "train"), (300, 3000, 7, 64, "search"),
                                           (200, 2500, 5, 256, "train"), (150, 1500, 5, 512, "search"),
                                           (90, 700, 4, 6, "train")])
def test_star_ops_against_oracle(N, T, R, D, order):
    """Bigger seeded graphs with a hub whose in-edge list is split into many chunks."""
    tri = synth(N, T, R, seed=N + D)
    build = G.build_train_graph if order == "train" else G.build_search_graph
    g_cpu = build(N, R, tri)
    s, d, _ = g_cpu.edges(form="all")
    og = OGraph(N, s, d, g_cpu.edata["e_type"], g_cpu.edata["norm"])
    g = g_cpu.to(DEV)
    assert g.plan()["n_hubs"] >= 1
    E = g.num_edges()
    gen = torch.Generator().manual_seed(D)
    x, x_in = torch.randn(E + N, D, generator=gen), torch.randn(E + N, D, generator=gen)
    gM, gN = torch.randn(E + N, D, generator=gen), torch.randn(N, D, generator=gen)
    for name in ("pre_mult", "pre_sub", "pre_add", "f_sparse_comp", "f_dense_comp", "f_comp", "a_sum", "a_mean", "a_max"):
        P = OO.init_params(name, D, gen)
        for k in P:
            if k.endswith("bias"):
                P[k] = torch.randn(P[k].shape, generator=gen) * 0.1
        gout = gN if name.startswith("a_") else gM
        a = x.clone().requires_grad_(True)
        b = x_in.clone().requires_grad_(True)
        Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        ref = OO.OPS[name](og, Pr, a, b)
        ref.backward(gout)
        op = O.MIXED_OPS[name]({"feature_dim": D, "drop_aggr": 0.0}).to(DEV)
        op.load_state_dict(P)
        out, ga, gb = run_module(op, g, x, x_in, gout)
        close(out, ref.detach(), f"{name} out")
        close(ga, a.grad, f"{name} ga")
        if b.grad is not None:
            close(gb, b.grad, f"{name} gb")
        for k, p in op.named_parameters():
            close(p.grad, Pr[k].grad, f"{name} grad {k}", rtol=3e-4, atol=1e-4)
    P = OO.init_params("f_sparse_last", D, gen)
    xn = torch.randn(N, D, generator=gen)
    a = xn.clone().requires_grad_(True)
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref = OO.f_sparse_last(og, Pr, a, a)
    ref.backward(gN)
    op = O.MIXED_OPS["f_sparse_last"]({"feature_dim": D}).to(DEV)
    op.load_state_dict(P)
    out, ga, _ = run_module(op, g, xn, xn, gN)
    close(out, ref.detach(), "f_sparse_last out")
    close(ga, a.grad, "f_sparse_last ga")
    for k, p in op.named_parameters():
        close(p.grad, Pr[k].grad, f"f_sparse_last grad {k}", rtol=3e-4, atol=1e-4)


def test_edge_cases_empty_and_isolated():
    D = 16
    # a graph with no edges at all: aggregators return the self rows, gates only see self rows
    g = G.RelGraph(5, torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long),
                   torch.zeros(0, dtype=torch.long), torch.zeros(0), device=DEV)
    x = torch.randn(5, D, device=DEV, requires_grad=True)
    for name in ("a_sum", "a_max", "a_mean"):
        op = O.MIXED_OPS[name]({"feature_dim": D, "drop_aggr": 0.0}).to(DEV)
        out = op(g, x, x)
        assert torch.equal(out, x.detach())
        out.sum().backward()
    op = O.MIXED_OPS["f_sparse_comp"]({"feature_dim": D}).to(DEV)
    out = op(g, x, x)
    assert out.shape == x.shape and torch.isfinite(out).all()
    # zero rows
    assert K.compose("sub", torch.zeros(0, D, device=DEV), torch.zeros(0, D, device=DEV)).shape == (0, D)


def test_linear_mfma_shapes():
    gen = torch.Generator().manual_seed(3)
    for rows, Kd, Nout in ((1, 8, 8), (130, 200, 200), (257, 64, 40), (1000, 100, 256), (300, 10, 7), (513, 256, 256), (77, 400, 200)):
        x = torch.randn(rows, Kd, generator=gen)
        W = torch.randn(Nout, Kd, generator=gen) / Kd ** 0.5
        b = torch.randn(Nout, generator=gen)
        gy = torch.randn(rows, Nout, generator=gen)
        for act in (None, "relu"):
            xr, Wr, br = (t.clone().requires_grad_(True) for t in (x, W, b))
            ref = torch.nn.functional.linear(xr.double(), Wr.double(), br.double())
            ref = torch.relu(ref) if act else ref
            ref.backward(gy.double())
            xd, Wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, W, b))
            out = K.linear(xd, Wd, bd, act)
            out.backward(gy.to(DEV))
            close(out, ref.float().detach(), f"linear {rows}x{Kd}x{Nout} {act}", rtol=2e-5, atol=2e-5)
            close(xd.grad, xr.grad, "linear gx", rtol=2e-5, atol=2e-5)
            close(Wd.grad, Wr.grad, "linear gW", rtol=5e-5, atol=5e-5)
            close(bd.grad, br.grad, "linear gb", rtol=5e-5, atol=5e-5)


def test_module_linear_matches_nn_linear():
    """functional.module_linear: an nn.Linear module on the library's row GEMM (entity projection, concat Linear) against the module
    itself in float64: output and the three gradients; anything but float32 HIP rows falls through to the module."""
    gen = torch.Generator().manual_seed(5)
    for rows, K_, Nout in ((14541, 100, 200), (3000, 1000, 200), (257, 64, 64)):
        mod = torch.nn.Linear(K_, Nout).to(DEV)
        x = torch.randn(rows, K_, generator=gen).to(DEV).requires_grad_(True)
        gy = torch.randn(rows, Nout, generator=gen).to(DEV)
        out = K.module_linear(mod, x)
        out.backward(gy)
        got = (out.detach(), x.grad.clone(), mod.weight.grad.clone(), mod.bias.grad.clone())
        mod.zero_grad()
        ref_mod = torch.nn.Linear(K_, Nout).double()
        ref_mod.load_state_dict({k: v.double().cpu() for k, v in mod.state_dict().items()})
        xr = x.detach().double().cpu().requires_grad_(True)
        ref = ref_mod(xr)
        ref.backward(gy.double().cpu())
        for g_, r_, what in zip(got, (ref.detach(), xr.grad, ref_mod.weight.grad, ref_mod.bias.grad), ("out", "gx", "gW", "gb")):
            err = float((g_.double().cpu() - r_).abs().max())
            assert err <= 1e-4 * max(1.0, float(r_.abs().max())), f"module_linear {rows}x{K_}x{Nout} {what}: {err:.3e}"
    mod = torch.nn.Linear(8, 4)
    assert torch.equal(K.module_linear(mod, torch.ones(3, 8)), mod(torch.ones(3, 8)))          # CPU rows: the module itself


def test_deferred_batchnorm_counters():
    """functional.deferred_counters: the BatchNorm step counters of the fused epilogues inside the block are bumped by ONE launch at
    its end (re-entrant), outside a block at once -- nn.BatchNorm1d's num_batches_tracked semantics either way."""
    rows, D = 500, 64
    bns = [torch.nn.BatchNorm1d(D).to(DEV) for _ in range(3)]
    ys = [torch.randn(rows, D, device=DEV) for _ in range(3)]
    w = torch.softmax(torch.randn(3, device=DEV), 0)
    K.mixed_epilogue(ys, bns, w)
    assert [int(b.num_batches_tracked) for b in bns] == [1, 1, 1]
    with K.deferred_counters():
        K.mixed_epilogue(ys, bns, w)
        with K.deferred_counters():
            K.mixed_epilogue(ys[:2], bns[:2], w[:2].contiguous())
        assert [int(b.num_batches_tracked) for b in bns] == [1, 1, 1]          # nothing yet: the outermost block flushes
    assert [int(b.num_batches_tracked) for b in bns] == [3, 3, 2]
    for b in bns:
        b.eval()
    K.mixed_epilogue(ys, bns, w)
    assert [int(b.num_batches_tracked) for b in bns] == [3, 3, 2]              # eval: no statistics, no count


def test_act_grad_transpose_kernel():
    """mrg_act_grad_transpose: gT[n][b] = g[b][n] * act'(y[b][n]) for the three activation codes, ragged tile edges, bit for bit
    against the torch expressions it replaces (same products in the same order)."""
    from mr_gnas_amd._lib import call, ptr, stream_of
    gen = torch.Generator().manual_seed(21)
    for B, N in ((1, 1), (70, 1000), (64, 64), (256, 5001), (3, 200000)):
        g = torch.randn(B, N, generator=gen).to(DEV)
        y = torch.sigmoid(torch.randn(B, N, generator=gen)).to(DEV)
        y[0, ::7] = 0.0
        for act, want in ((0, g), (1, g * (y > 0)), (2, g * y * (1 - y))):
            out = torch.full((N, B), float("nan"), device=DEV)
            call("mrg_act_grad_transpose", (ptr(g), ptr(y) if act else None, ptr(out), B, N, act, stream_of(g)))
            assert torch.equal(out, want.t().contiguous()), (B, N, act)
    lib = mr_gnas_amd._lib.load()
    assert lib.mrg_act_grad_transpose(None, None, None, 4, 4, 0, None) == -1           # MRG_E_NULLPTR
    assert lib.mrg_act_grad_transpose(None, None, None, 4, 4, 7, None) == -3           # MRG_E_ENUM
    assert lib.mrg_act_grad_transpose(None, None, None, 0, 4, 0, None) == 0


def test_linear_wide_short_gradients():
    """[B, N] scores against an entity table (Nout = N >> rows): the input gradient is a reduction over the N entity rows and
    runs on the split-over-rows weight-gradient kernel (functional._Linear.backward), the row GEMM where that kernel does
    not take the shape; both against float64 and against each other."""
    gen = torch.Generator().manual_seed(11)
    for rows, Kd, Nout in ((90, 200, 3000), (256, 200, 5001), (1000, 64, 2048), (1001, 64, 2048), (256, 50, 3000), (4, 256, 20000)):
        x = torch.randn(rows, Kd, generator=gen) * 0.3
        W = torch.randn(Nout, Kd, generator=gen) * 0.3
        gy = torch.randn(rows, Nout, generator=gen)
        xr, Wr = x.double().requires_grad_(True), W.double().requires_grad_(True)
        torch.sigmoid(xr @ Wr.t()).backward(gy.double())
        got = {}
        for sw in (True, False):
            old, K.switches.WIDE_BWD_INPUT = K.switches.WIDE_BWD_INPUT, sw
            try:
                xd, Wd = x.to(DEV).requires_grad_(True), W.to(DEV).requires_grad_(True)
                K.linear(xd, Wd, None, "sigmoid").backward(gy.to(DEV))
            finally:
                K.switches.WIDE_BWD_INPUT = old
            got[sw] = xd.grad
            for g_, r_, what in ((xd.grad, xr.grad, "gx"), (Wd.grad, Wr.grad, "gW")):
                err = float((g_.double().cpu() - r_).abs().max())
                assert err <= 1e-4 * max(1.0, float(r_.abs().max())), f"wide linear {rows}x{Kd}x{Nout} split={sw} {what}: {err:.3e}"
        assert float((got[True] - got[False]).abs().max()) <= 1e-4 * max(1.0, float(xr.grad.abs().max()))


@pytest.mark.parametrize("rows,D,present", [(3000, 200, [0, 1, 1, 1, 1]), (777, 64, [1, 1, 1]), (100, 10, [1, 0, 1, 1]),
                                            (5000, 256, [1, 1, 1, 1, 1, 1, 1, 1]), (64, 512, [1, 1])])
def test_mixed_epilogue_matches_bn_relu_sum(rows, D, present):
    """Fused MixedOp epilogue vs the reference formulation sum_k w_k relu(BatchNorm1d_k(y_k))
    (reference models/cell_lp.py:25-33) evaluated by torch in float64 on the CPU."""
    gen = torch.Generator().manual_seed(rows + D)
    Kb = len(present)
    ys = [torch.randn(rows, D, generator=gen) * (1 + k) + 0.5 * k if p else None for k, p in enumerate(present)]
    gam = [torch.rand(D, generator=gen) + 0.5 for _ in range(Kb)]
    bet = [torch.randn(D, generator=gen) * 0.3 for _ in range(Kb)]
    w = torch.softmax(torch.randn(Kb, generator=gen), 0)
    gout = torch.randn(rows, D, generator=gen)
    # reference (float64, torch BN in training mode)
    yr = [(y if y is not None else torch.zeros(rows, D)).double().requires_grad_(True) for y in ys]
    gr_, br_, wr = [t.double().requires_grad_(True) for t in gam], [t.double().requires_grad_(True) for t in bet], w.double().requires_grad_(True)
    ref = sum(wr[k] * torch.relu(torch.nn.functional.batch_norm(yr[k], None, None, gr_[k], br_[k], training=True)) for k in range(Kb))
    ref.backward(gout.double())
    # HIP
    bns = [torch.nn.BatchNorm1d(D).to(DEV) for _ in range(Kb)]
    for k, b in enumerate(bns):
        b.weight.data.copy_(gam[k]); b.bias.data.copy_(bet[k]); b.train()
    yd = [y.to(DEV).requires_grad_(True) if y is not None else None for y in ys]
    wd = w.to(DEV).requires_grad_(True)
    out = K.mixed_epilogue(yd, bns, wd)
    out.backward(gout.to(DEV))
    close(out, ref.float().detach(), "mixed out", rtol=2e-5, atol=2e-5)
    close(wd.grad, wr.grad.float(), "mixed dw", rtol=1e-4, atol=1e-3)
    for k in range(Kb):
        close(bns[k].weight.grad, gr_[k].grad.float(), f"dgamma {k}", rtol=1e-4, atol=2e-4)
        close(bns[k].bias.grad, br_[k].grad.float(), f"dbeta {k}", rtol=1e-4, atol=2e-4)
        if ys[k] is not None:
            close(yd[k].grad, yr[k].grad.float(), f"dy {k}", rtol=1e-4, atol=2e-6)
            ref_bn = torch.nn.BatchNorm1d(D).double()
            ref_bn.train(); ref_bn(ys[k].double())
            close(bns[k].running_mean, ref_bn.running_mean.float(), f"running_mean {k}", rtol=1e-5, atol=1e-6)
            close(bns[k].running_var, ref_bn.running_var.float(), f"running_var {k}", rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("N,R,T,D", [(50, 7, 400, 32), (300, 11, 5000, 200), (40, 3, 0, 16), (64, 5, 777, 10)])
def test_distmult_score_and_gradients(N, R, T, D):
    """Fused DistMult (reference models/model_search_lp.py:169-176) against the float64 definition;
    empty batch, ragged D, entities / relations that never occur (zero gradient rows)."""
    gen = torch.Generator().manual_seed(N + T)
    ent = torch.randn(N, D, generator=gen)
    rel = torch.randn(R, D, generator=gen)
    trip = torch.stack((torch.randint(0, N - 3, (T,), generator=gen), torch.randint(0, R - 1, (T,), generator=gen),
                        torch.randint(0, N - 3, (T,), generator=gen)), dim=1)
    w = torch.randn(T, generator=gen)
    e64, r64 = ent.double().requires_grad_(), rel.double().requires_grad_()
    ref = torch.sum(e64[trip[:, 0]] * r64[trip[:, 1]] * e64[trip[:, 2]], dim=1)
    (ref * w.double()).sum().backward()
    ed, rd = ent.cuda().requires_grad_(), rel.cuda().requires_grad_()
    sp = K.ScorePlan(trip.cuda(), N, R)
    got = K.distmult_score(ed, rd, sp)
    (got * w.cuda()).sum().backward()
    assert got.shape == (T,)
    scale = max(1.0, float(ref.detach().abs().max()) if T else 1.0)
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().float().numpy(), atol=1e-4 * scale, rtol=1e-4)
    for a, b in ((ed.grad, e64.grad), (rd.grad, r64.grad)):
        s = max(1.0, float(b.abs().max()))
        np.testing.assert_allclose(a.cpu().numpy(), b.float().numpy(), atol=1e-4 * s, rtol=1e-4)
    assert torch.all(ed.grad[N - 3:] == 0) and torch.all(rd.grad[R - 1:] == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,K1,K2,Nout", [(70000, 200, 200, 200), (70001, 200, 0, 200), (3000, 100, 100, 100), (513, 64, 0, 40),
                                             (66000, 52, 0, 300), (259, 400, 0, 7), (272115, 200, 0, 200), (65537, 128, 0, 450), (300001, 128, 128, 128), (200001, 256, 0, 256)])
def test_split_core_is_as_accurate_as_the_exact_f32_core(rows, K1, K2, Nout):
    """The split-bf16 matrix core (six bf16 cross terms, f32 accumulate) against the exact-f32 MFMA core, both
    measured against a float64 product: its error may not exceed 1.5x the exact core's (+ 1e-6 of the output
    scale).  Covers dual-source K, K % 16 != 0, ragged row blocks, >1 column block, both row-tile shapes."""
    lib = mr_gnas_amd._lib.load()
    gen = torch.Generator().manual_seed(rows + Nout)
    s = (torch.randn(rows, K1, generator=gen) * 3).to(DEV)
    s_in = torch.randn(rows, K2, generator=gen).to(DEV) if K2 else None
    W = (torch.randn(Nout, K1 + K2, generator=gen) / (K1 + K2) ** 0.5).to(DEV)
    b = torch.randn(Nout, generator=gen).to(DEV)
    x = s if s_in is None else torch.cat((s, s_in), 1)
    ref = torch.nn.functional.linear(x.double(), W.double(), b.double())
    gy = torch.randn(rows, Nout, generator=gen).to(DEV)
    ref_gx = gy.double() @ W.double()
    errs = {}
    try:
        for mode in (0, 1, 2):                      # 2: the one-wave kernel (opt-in comparison point)
            assert lib.mrg_gemm_set_mode(mode) == 0
            if K2 == 0:
                out = K.linear(x, W, b, None)
                xg = x.clone().requires_grad_(True)
                K.linear(xg, W, b, None).backward(gy)
                errs[mode] = (float((out.double() - ref).abs().max()), float((xg.grad.double() - ref_gx).abs().max()))
            else:
                out = torch.empty(rows, Nout, device=DEV)
                ws = torch.empty(int(lib.mrg_gemm_workspace_bytes(K1 + K2, Nout)), dtype=torch.uint8, device=DEV)
                from mr_gnas_amd._lib import call, ptr, stream_of
                call("mrg_dense_filter_fwd", (1, ptr(s), ptr(s_in), ptr(W), ptr(b), None, 1.0, ptr(out), None, ptr(ws), rows, K1,
                                              stream_of(out)))
                errs[mode] = (float((out.double() - ref).abs().max()), 0.0)
    finally:
        lib.mrg_gemm_set_mode(0)
    for i, scale in ((0, float(ref.abs().max())), (1, float(ref_gx.abs().max()))):
        for mode in (0, 2):
            assert errs[mode][i] <= 1.5 * errs[1][i] + 1e-6 * scale, (mode, errs, scale)
            assert errs[mode][i] <= 2e-5 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("rows,K1,K2,Nout", [(70001, 200, 200, 200), (40000, 200, 0, 200), (3000, 100, 100, 100), (517, 64, 0, 40),
                                             (20000, 52, 0, 128), (259, 400, 0, 8), (16, 200, 200, 200), (0, 64, 64, 64),
                                             (5000, 256, 0, 256), (3001, 128, 128, 300), (777, 512, 0, 512)])
def test_split_core_weight_gradient(rows, K1, K2, Nout):
    """gW = gY^T [X1 | X2], gb = column sums of gY on the split-bf16 core against the exact-f32 core, both against
    float64: ragged last row tile, 1 / 2 row-tile groups, 1..2 column blocks, bias column, empty input."""
    from mr_gnas_amd._lib import call, ptr, stream_of
    lib = mr_gnas_amd._lib.load()
    gen = torch.Generator().manual_seed(rows + Nout + K2)
    gy = torch.randn(rows, Nout, generator=gen).to(DEV)
    x1 = (torch.randn(rows, K1, generator=gen) * 2).to(DEV)
    x2 = torch.randn(rows, K2, generator=gen).to(DEV) if K2 else None
    x = x1 if x2 is None else torch.cat((x1, x2), 1)
    ref_w = gy.double().t() @ x.double()
    ref_b = gy.double().sum(0)
    errs = {}
    try:
        for mode in (0, 1):
            assert lib.mrg_gemm_set_mode(mode) == 0
            gW, gb = torch.full((Nout, K1 + K2), 7.0, device=DEV), torch.full((Nout,), 7.0, device=DEV)
            ws = torch.empty(max(16, int(lib.mrg_linear_bwd_weight_workspace_bytes(rows, K1 + K2, Nout))), dtype=torch.uint8, device=DEV)
            call("mrg_linear_bwd_weight", (ptr(gy), ptr(x1), ptr(x2), ptr(gW), ptr(gb), ptr(ws), rows, K1, K2, Nout, stream_of(gW)))
            errs[mode] = (float((gW.double() - ref_w).abs().max()), float((gb.double() - ref_b).abs().max()))
    finally:
        lib.mrg_gemm_set_mode(0)
    for i, scale in ((0, max(1.0, float(ref_w.abs().max()) if rows else 1.0)), (1, max(1.0, float(ref_b.abs().max()) if rows else 1.0))):
        assert errs[0][i] <= 1.5 * errs[1][i] + 2e-6 * scale, (errs, scale)
        assert errs[0][i] <= 2e-5 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("rows,K1,K2,Nout", [(70001, 200, 200, 200), (40000, 200, 0, 200), (3000, 100, 100, 100), (517, 64, 0, 40),
                                             (31, 200, 200, 200), (16, 64, 64, 64), (5000, 256, 0, 256), (3001, 128, 128, 300), (33, 52, 0, 128),
                                             (4800, 200, 200, 200), (9600, 200, 0, 200)])
def test_weight_gradient_split_once_is_bit_exact_with_split_per_wave(rows, K1, K2, Nout):
    """wgrad_x3v_k (every operand fragment split into its bf16 planes once per workgroup, shared through LDS) against
    wgrad_x3_k (split by every wave that multiplies it): same operands, same products, same order -- gW and gb bit-identical for
    any row count (a ragged last tile holds its valid rows first and zeros after in both kernels), for the single row range
    (mrg_linear_bwd_weight) and the three direction segments in one launch (mrg_linear_bwd_weight3)."""
    from mr_gnas_amd._lib import call, ptr, ptr_array, stream_of
    lib = mr_gnas_amd._lib.load()
    gen = torch.Generator().manual_seed(rows + Nout + K2 + 1)
    gy = torch.randn(rows, Nout, generator=gen).to(DEV)
    x1 = (torch.randn(rows, K1, generator=gen) * 2).to(DEV)
    x2 = torch.randn(rows, K2, generator=gen).to(DEV) if K2 else None
    b0, b1 = rows // 3, rows - rows // 4
    res = {}
    try:
        for variant in (1, 0):
            assert lib.mrg_wgrad_set_variant(variant) == 0
            gW, gb = torch.full((Nout, K1 + K2), 7.0, device=DEV), torch.full((Nout,), 7.0, device=DEV)
            ws = torch.empty(max(16, int(lib.mrg_linear_bwd_weight_workspace_bytes(rows, K1 + K2, Nout))), dtype=torch.uint8, device=DEV)
            call("mrg_linear_bwd_weight", (ptr(gy), ptr(x1), ptr(x2), ptr(gW), ptr(gb), ptr(ws), rows, K1, K2, Nout, stream_of(gW)))
            out = [gW, gb]
            ws3 = int(lib.mrg_linear_bwd_weight3_workspace_bytes(b0, b1, rows, K1, K2, Nout))
            if ws3 > 0:
                gWs = [torch.full((Nout, K1 + K2), 7.0, device=DEV) for _ in range(3)]
                gbs = [torch.full((Nout,), 7.0, device=DEV) for _ in range(3)]
                w3 = torch.empty(ws3, dtype=torch.uint8, device=DEV)
                call("mrg_linear_bwd_weight3", (ptr(gy), ptr(x1), ptr(x2), ptr_array(gWs), ptr_array(gbs), ptr(w3), b0, b1, rows, K1, K2, Nout, stream_of(gW)))
                out += gWs + gbs
            torch.cuda.synchronize()
            res[variant] = out
    finally:
        lib.mrg_wgrad_set_variant(1)
    assert len(res[0]) == len(res[1])
    for i, (a, b) in enumerate(zip(res[1], res[0])):
        assert torch.equal(a, b), f"output {i}: max diff {float((a - b).abs().max())}"
    x = x1 if x2 is None else torch.cat((x1, x2), 1)
    ref = gy.double().t() @ x.double()
    assert float((res[1][0].double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("N,E,D,K_,self_rows", [(300, 5000, 200, 3, True), (50, 0, 64, 2, True), (1000, 70000, 100, 0, True), (200, 3000, 52, 8, False),
                                              (77, 900, 260, 1, True)])
def test_fan_in_sum_with_a_gathered_term(N, E, D, K_, self_rows):
    """mrg_sum_rows_gather: sum of K [M, D] tensors plus a gather of [N, D] rows (edge row e <- dst[e], self row n <- n) against
    torch; then a_sum's input gradient left to the fan-in sum (functional.switches.LAZY_ASUM) against the materialised form through a Fan."""
    from mr_gnas_amd._lib import call, ptr, ptr_array, stream_of
    gen = torch.Generator().manual_seed(N + E + D + K_)
    dst = torch.randint(0, N, (E,), generator=gen)
    xs = [torch.randn(E + N, D, generator=gen).to(DEV) for _ in range(K_)]
    ge, gs_ = torch.randn(N, D, generator=gen).to(DEV), torch.randn(N, D, generator=gen).to(DEV)
    out = torch.full((E + N, D), 7.0, device=DEV)
    call("mrg_sum_rows_gather", (ptr_array(xs), K_, ptr(ge), ptr(gs_) if self_rows else None, ptr(dst.to(DEV).int()), E, E + N, D, ptr(out), stream_of(out)))
    ref = torch.cat((ge[dst.to(DEV)], gs_ if self_rows else torch.zeros(N, D, device=DEV)))
    for x in xs:
        ref = ref + x
    close(out, ref.cpu(), "sum with a gathered term", rtol=1e-6, atol=1e-6)
    if E == 0 or D > 256:
        return
    g = G.RelGraph(N, torch.randint(0, N, (E,), generator=gen).numpy(), dst.numpy(), torch.randint(0, 4, (E,), generator=gen).numpy(),
                   (torch.rand(E, generator=gen) + 0.1).numpy().astype(np.float32), device=DEV)
    x0 = torch.randn(E + N, D, generator=gen)
    gout = torch.randn(N, D, generator=gen).to(DEV)
    res = {}
    try:
        for lazy in (True, False):
            K.switches.LAZY_ASUM = lazy
            x = x0.clone().to(DEV).requires_grad_(True)
            fan = K.Fan(x, 3)
            y = K.aggregate_rows("sum", fan.take(), g) + K.aggregate_rows("mean", fan.take(), g)
            (y * gout).sum().backward()
            res[lazy] = x.grad.clone()
    finally:
        K.switches.LAZY_ASUM = True
    close(res[True], res[False].cpu(), "a_sum gradient through the fan-in gather", rtol=2e-6, atol=2e-6)


def test_compose_broadcast_relation_row():
    """Advisor r1: the reference's pre-ops broadcast (`src_emb - hr` with hr [1, D]); the gradient of a
    broadcast hr must come back in hr's own shape."""
    gen = torch.Generator().manual_seed(3)
    s = torch.randn(50, 16, generator=gen)
    hr = torch.randn(1, 16, generator=gen)
    g_up = torch.randn(50, 16, generator=gen)
    for kind, f in (("sub", lambda a, b: a - b), ("mult", lambda a, b: a * b), ("add", lambda a, b: a + b)):
        a, b = s.clone().requires_grad_(True), hr.clone().requires_grad_(True)
        f(a, b).backward(g_up)
        ad, bd = s.to(DEV).requires_grad_(True), hr.to(DEV).requires_grad_(True)
        out = K.compose(kind, ad, bd)
        out.backward(g_up.to(DEV))
        assert bd.grad.shape == hr.shape
        close(out, f(s, hr), f"compose {kind} broadcast out")
        close(ad.grad, a.grad, f"compose {kind} broadcast gs")
        close(bd.grad, b.grad, f"compose {kind} broadcast ghr", rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("N,T,R,D,order", [(400, 5000, 9, 200, "search"), (120, 900, 4, 64, "train"), (90, 700, 4, 6, "train")])
def test_filters_with_both_operands_the_same_rows(N, T, R, D, order):
    """The first two MixedOps of a cell call every FIRST operator as op(g, h_in, h_in) (reference
    models/cell_lp.py:95-104).  The HIP operators then fold the weight halves, W [s ; s] = (W[:, :D] + W[:, D:]) s, and
    return the whole input gradient through the first operand; against the oracle's literal cat([s, s]) formulation,
    once with the very same tensor object and once with two aliases of one storage (what functional.Fan hands out)."""
    tri = synth(N, T, R, seed=N + D)
    build = G.build_train_graph if order == "train" else G.build_search_graph
    g_cpu = build(N, R, tri)
    s, d, _ = g_cpu.edges(form="all")
    og = OGraph(N, s, d, g_cpu.edata["e_type"], g_cpu.edata["norm"])
    g = g_cpu.to(DEV)
    E = g.num_edges()
    gen = torch.Generator().manual_seed(D + 1)
    x, gM = torch.randn(E + N, D, generator=gen), torch.randn(E + N, D, generator=gen)
    for name in ("f_dense_comp", "f_sparse_comp", "f_comp", "f_dense", "f_sparse"):
        P = OO.init_params(name, D, gen)
        for k in P:
            if k.endswith("bias"):
                P[k] = torch.randn(P[k].shape, generator=gen) * 0.1
        a = x.clone().requires_grad_(True)
        Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        ref = OO.OPS[name](og, Pr, a, a)
        ref.backward(gM)
        for how in ("same object", "aliases"):
            op = O.MIXED_OPS[name]({"feature_dim": D, "drop_aggr": 0.0}).to(DEV)
            op.load_state_dict(P)
            xd = x.to(DEV).requires_grad_(True)
            u, v = (xd, xd) if how == "same object" else (xd.view_as(xd), xd.view_as(xd))
            assert K.same_rows(u, v)
            out = op(g, u, v)
            out.backward(gM.to(DEV))
            close(out, ref.detach(), f"{name} ({how}) out")
            close(xd.grad, a.grad, f"{name} ({how}) input grad")
            for k, p in op.named_parameters():
                close(p.grad, Pr[k].grad, f"{name} ({how}) grad {k}", rtol=3e-4, atol=1e-4)
    assert not K.same_rows(x, gM) and not K.same_rows(x[:10], x[1:11])


@pytest.mark.parametrize("E,nseg,D", [(400_000, 7, 200), (3_000_000, 474, 32), (60_000, 3, 36)])
def test_span_sum_with_few_long_segments(E, nseg, D):
    """Segments that are few and very long (the relation gradient of DistMult: 3 M triples over 474 rows): the plan
    lengthens its spans (graph.auto_span) and the hub pass runs the wide workgroup; checked against a float64 sum,
    and bitwise reproducible from one launch to the next."""
    gen = torch.Generator().manual_seed(E + nseg)
    seg = torch.randint(0, nseg, (E,), generator=gen)
    seg[: E // 2] = 1                                                     # one segment holds half of the elements
    X = torch.randn(5000, D, generator=gen)
    xi = torch.randint(0, 5000, (E,), generator=gen)
    plan = G.span_plan(seg.to(DEV), nseg)
    assert plan["span"] == G.auto_span(E, nseg) and (E < 3_000_000 or plan["span"] > G.SPAN_ELEMS)
    meta = G.span_meta(plan, xi.to(DEV))
    out = K.span_gcs("copy", X.to(DEV), None, meta, plan)
    ref = torch.zeros(nseg, D, dtype=torch.float64).index_add_(0, seg, X.double()[xi])
    close(out, ref.float(), "span sum, long segments", rtol=2e-6 * (E / nseg) ** 0.5)
    assert torch.equal(out, K.span_gcs("copy", X.to(DEV), None, meta, plan))


@pytest.mark.parametrize("N,E,R,D,hub", [(3000, 70000, 11, 200, False), (500, 90000, 7, 64, True), (4000, 20000, 5, 128, False), (64, 300, 3, 52, True),
                                         (700, 30000, 5, 96, False), (900, 25000, 5, 192, True)])
def test_fused_amax_is_bit_exact_with_the_two_launch_form(N, E, R, D, hub):
    """a_max as one GEMM with the ReLU + segmented-max epilogue (mrg_linear_relu_segmax_fwd) against linear -> ReLU ->
    segmented max as separate launches: outputs, argmax routing (ties included: half of the ReLU outputs are exact zeros) and
    every gradient must be bit-identical; nodes without in-edges, a hub that spans many row tiles."""
    gen = torch.Generator().manual_seed(N + E)
    src = torch.randint(0, N, (E,), generator=gen)
    dst = torch.randint(0, N - 10, (E,), generator=gen)              # the last 10 nodes have no in-edge
    if hub:
        dst[: E // 2] = 3
    et = torch.randint(0, R, (E,), generator=gen)
    g = G.RelGraph(N, src.numpy(), dst.numpy(), et.numpy(), np.ones(E, np.float32), device=DEV)
    x0 = torch.randn(E + N, D, generator=gen)
    W0 = torch.randn(D, D, generator=gen) / D ** 0.5
    b0 = torch.randn(D, generator=gen) * 0.1
    gout = torch.randn(N, D, generator=gen).to(DEV)
    res = {}
    min_rows = K.switches.FUSED_AMAX_MIN_ROWS
    try:
        K.switches.FUSED_AMAX_MIN_ROWS = 0
        for fused in (True, False, "sparse"):
            K.switches.FUSED_AMAX = bool(fused)
            K.switches.SPARSE_AMAX_BWD = fused == "sparse"           # the bit comparison is between the two DENSE input-gradient products
            x, W, b = (t.clone().to(DEV).requires_grad_(True) for t in (x0, W0, b0))
            out = K.linear_relu_aggregate("max", x, W, b, g)
            out.backward(gout)
            res[fused] = (out.detach(), x.grad, W.grad, b.grad)
    finally:
        K.switches.FUSED_AMAX, K.switches.FUSED_AMAX_MIN_ROWS, K.switches.SPARSE_AMAX_BWD = True, min_rows, True
    assert int(mr_gnas_amd._lib.load().mrg_linear_relu_segmax_workspace_bytes(N, D, D)) > 0
    for a, b_, what in zip(res[True], res[False], ("out", "gx", "gW", "gb")):
        assert torch.equal(a, b_), what
    # the input gradient without a dense product (mrg_segmax_bwd_input: the rows of W of the columns an edge won, exact f32): the same
    # message gradient bit for bit (so gW, gb too), gx within rounding of the split-core product -- and closer to float64 than it
    assert mr_gnas_amd._lib.load().mrg_segmax_bwd_input_ok(D, D) == 1
    for i, what in ((0, "out"), (2, "gW"), (3, "gb")):
        assert torch.equal(res["sparse"][i], res[True][i]), what
    assert torch.equal(res["sparse"][1][E:], res[True][1][E:])
    close(res["sparse"][1], res[True][1].cpu(), "sparse a_max input gradient vs the dense product", rtol=2e-5, atol=2e-6, rms_rtol=2e-5)
    gy64 = torch.zeros(E, D, dtype=torch.float64, device=DEV)
    xg = x0.to(DEV)
    m64 = torch.relu(xg[:E].double() @ W0.to(DEV).double().t() + b0.to(DEV).double())
    dstd = dst.to(DEV)
    top = torch.zeros(N, D, dtype=torch.float64, device=DEV).scatter_reduce(0, dstd.view(-1, 1).expand(E, D), m64, "amax", include_self=False)
    win = (m64 == top[dstd]) & (m64 > 0)
    first = torch.full((N, D), E, dtype=torch.long, device=DEV).scatter_reduce(0, dstd.view(-1, 1).expand(E, D),
                                                                               torch.where(win, torch.arange(E, device=DEV).view(-1, 1).expand(E, D), E), "amin")
    gy64 = torch.where(win & (first[dstd] == torch.arange(E, device=DEV).view(-1, 1)), gout.double()[dstd], 0.0)
    gx64 = gy64 @ W0.to(DEV).double()
    e_sparse = float((res["sparse"][1][:E].double() - gx64).abs().max())
    e_dense = float((res[True][1][:E].double() - gx64).abs().max())
    assert e_sparse <= max(2.0 * e_dense, 1e-6 * float(gx64.abs().max())), f"sparse {e_sparse:.3e} vs dense {e_dense:.3e} against float64"
    ref = OO.a_max(OGraph(N, src.numpy(), dst.numpy(), et.numpy(), np.ones(E, np.float32)), {"linear.weight": W0, "linear.bias": b0}, x0, None)
    close(res[True][0], ref, "fused a_max vs oracle")


@pytest.mark.parametrize("kind,tied", [("f_dense_comp", False), ("f_comp", False), ("f_dense_comp", True), ("f_comp", True)])
@pytest.mark.parametrize("N,E,R,D", [(2000, 90000, 9, 200), (300, 5000, 4, 64), (50, 0, 2, 64)])
def test_grouped_direction_segments_are_bit_exact(kind, tied, N, E, R, D):
    """The dense filters with their three direction segments in one launch each (mrg_dense_filter_fwd3 / ..._bwd3 entry
    points) against one launch per segment: outputs and every gradient bit-identical (same per-row arithmetic); ragged
    segment boundaries, an empty graph (self rows only)."""
    gen = torch.Generator().manual_seed(N + E + D)
    src = torch.randint(0, N, (E,), generator=gen)
    dst = torch.randint(0, N, (E,), generator=gen)
    et = torch.randint(0, 2 * R, (E,), generator=gen)
    g = G.RelGraph(N, src.numpy(), dst.numpy(), et.numpy(), (torch.rand(E, generator=gen) + 0.1).numpy().astype(np.float32), device=DEV)
    op = O.MIXED_OPS[kind]({"feature_dim": D}).to(DEV)
    a0 = torch.randn(E + N, D, generator=gen)
    b0 = a0 if tied else torch.randn(E + N, D, generator=gen)
    gout = torch.randn(E + N, D, generator=gen).to(DEV)
    res = {}
    try:
        for grouped in (True, False):
            K.switches.GROUPED_SEGMENTS = grouped
            a = a0.clone().to(DEV).requires_grad_(True)
            b = a if tied else b0.clone().to(DEV).requires_grad_(True)
            op.zero_grad()
            out = op(g, a, b)
            out.backward(gout)
            res[grouped] = [out.detach(), a.grad] + ([] if tied else [b.grad]) + [p.grad.clone() for p in op.parameters()]
    finally:
        K.switches.GROUPED_SEGMENTS = True
    for x, y in zip(res[True], res[False]):
        assert torch.equal(x, y)


@pytest.mark.parametrize("rows,D,K_", [(70001, 200, 4), (513, 64, 3), (5, 36, 2)])
def test_fused_statistics_to_coefficients_is_bit_exact(rows, D, K_):
    """mrg_mix_stats_coef (statistics kernel, then reduction + finalize in one launch) against mrg_mix_colstats +
    mrg_mix_finalize_fwd: coefficients and running statistics bit-identical."""
    from mr_gnas_amd._lib import call, ptr, ptr_array, stream_of
    lib = mr_gnas_amd._lib.load()
    gen = torch.Generator().manual_seed(rows + D)
    ys = [torch.randn(rows, D, generator=gen).to(DEV) * (k + 1) for k in range(K_)]
    ys[1] = None                                                      # an all-zero branch (f_zero)
    gam = [torch.rand(D, generator=gen).to(DEV) for _ in range(K_)]
    bet = [torch.randn(D, generator=gen).to(DEV) for _ in range(K_)]
    out = {}
    for fused in (True, False):
        rm = [torch.zeros(D, device=DEV) for _ in range(K_)]
        rv = [torch.ones(D, device=DEV) for _ in range(K_)]
        coef = torch.full((K_, 4, D), 7.0, device=DEV)
        ws = torch.empty(int(lib.mrg_mix_workspace_bytes(K_, D)), dtype=torch.uint8, device=DEV)
        st = stream_of(coef)
        if fused:
            call("mrg_mix_stats_coef", (ptr_array(ys), ptr_array(gam), ptr_array(bet), ptr_array(rm), ptr_array(rv), K_, rows, float(rows), D, 1e-5, 0.1,
                                        ptr(coef), ptr(ws), None, st))
        else:
            sums = torch.empty(K_, 2, D, dtype=torch.float64, device=DEV)
            call("mrg_mix_colstats", (ptr_array(ys), K_, rows, D, ptr(sums), ptr(ws), None, st))
            call("mrg_mix_finalize_fwd", (ptr(sums), ptr_array(gam), ptr_array(bet), ptr_array(rm), ptr_array(rv), K_, float(rows), D, 1e-5, 0.1, ptr(coef), st))
        out[fused] = [coef] + rm + rv
    for a, b in zip(out[True], out[False]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("N,E,R,D,hub", [(3000, 70000, 11, 200, False), (500, 90000, 7, 64, True), (4000, 20000, 5, 128, False), (64, 300, 3, 52, True),
                                         (40, 33, 2, 100, False),
                                         # column-tile counts that gemm_pick_nt pads (3 -> 4, 5 / 6 -> 7): the padding tile has no ReLU-mask
                                         # word (advisor r2: a stray store zeroed the next edge row's word 0)
                                         (700, 30000, 5, 96, False), (700, 30000, 5, 160, True), (900, 25000, 5, 192, False), (300, 9000, 3, 68, False)])
def test_fused_amean_matches_the_two_launch_form(N, E, R, D, hub):
    """a_mean as GEMM with the run-sum epilogue + heads reducer + bit-mask backward (no [E, D] messages) against linear ->
    ReLU -> span reducer: outputs and gradients within float summation-order tolerance (the association differs), the ReLU
    bit mask exact (checked through the input gradient of rows whose message is dead), bitwise reproducible; a hub that spans
    many strips, nodes without in-edges, a list shorter than a strip half."""
    gen = torch.Generator().manual_seed(N + E + 1)
    src = torch.randint(0, N, (E,), generator=gen)
    dst = torch.randint(0, N - 10, (E,), generator=gen)
    if hub:
        dst[: E // 2] = 3
    et = torch.randint(0, R, (E,), generator=gen)
    g = G.RelGraph(N, src.numpy(), dst.numpy(), et.numpy(), np.ones(E, np.float32), device=DEV)
    x0 = torch.randn(E + N, D, generator=gen)
    W0 = torch.randn(D, D, generator=gen) / D ** 0.5
    b0 = torch.randn(D, generator=gen) * 0.1
    gout = torch.randn(N, D, generator=gen).to(DEV)
    res = {}
    min_rows = K.switches.FUSED_AMAX_MIN_ROWS
    try:
        K.switches.FUSED_AMAX_MIN_ROWS = 0
        for fused in (True, True, False):
            K.switches.FUSED_AMEAN = fused
            x, W, b = (t.clone().to(DEV).requires_grad_(True) for t in (x0, W0, b0))
            out = K.linear_relu_aggregate("mean", x, W, b, g)
            out.backward(gout)
            cur = (out.detach(), x.grad, W.grad, b.grad)
            if fused and True in res:
                assert all(torch.equal(p, q) for p, q in zip(cur, res[True])), "fused a_mean is not reproducible"
            res[fused] = cur
    finally:
        K.switches.FUSED_AMEAN, K.switches.FUSED_AMAX_MIN_ROWS = True, min_rows
    for a, b_, what in zip(res[True], res[False], ("out", "gx", "gW", "gb")):
        close(a, b_.cpu(), "fused a_mean " + what, rtol=2e-5, atol=1e-6)
    # exact mask: rows whose message is dead in every column get a zero input gradient in both forms, and vice versa
    dead_f, dead_u = (res[True][1][:E] == 0), (res[False][1][:E] == 0)
    assert torch.equal(dead_f.all(1), dead_u.all(1))
    # ... and per 32-column block of the mask words: the bias gradient sums gy * mask over all edges, so a zeroed word shows
    gy_f = (res[True][3] - res[False][3]).abs().max()
    assert float(gy_f) <= 2e-5 * max(1.0, float(res[False][3].abs().max()))
    ref = OO.a_mean(OGraph(N, src.numpy(), dst.numpy(), et.numpy(), np.ones(E, np.float32)), {"linear.weight": W0, "linear.bias": b0}, x0, None)
    close(res[True][0], ref, "fused a_mean vs oracle")


@pytest.mark.gpu
@pytest.mark.parametrize("rows,K1,K2,Nout", [(70001, 200, 0, 200), (66000, 200, 200, 200), (513, 64, 0, 40), (259, 400, 0, 8), (65537, 128, 0, 452),
                                             (300001, 128, 0, 128), (33, 200, 0, 200), (20000, 96, 0, 96), (9000, 52, 0, 300), (131073, 256, 0, 256)])
def test_split_core_kernels_are_bit_exact_with_each_other(rows, K1, K2, Nout):
    """The four forms of the split-core row GEMM compute the same sums in the same order: the default kernel with the
    weight slabs shared through LDS (mode 0, gemm_x3s.hpp) with accumulator-order stores (the default) and on transposed
    accumulators with 16-byte epilogue accesses (mrg_gemm_set_epilogue(2), round 4's comparison point), the wave-autonomous kernel
    (mode 2) with accumulator-order stores and with row-order 16-byte stores through LDS (mrg_gemm_set_epilogue(1)).  Bias / ReLU / sigmoid, gate (+ stored
    gate), row scale, accumulate -- bit-identical outputs; ragged last strips, partial last column tiles, two column blocks,
    both row-tile shapes, dual-source K.  Third switch: the eight-tile column block of a 256-wide output (gemm_x3s8.hpp, the
    default for 224 < N <= 256) against the 2 x 4-tile kernel (mrg_gemm_set_wide8(0))."""
    from mr_gnas_amd._lib import call, ptr, stream_of
    lib = mr_gnas_amd._lib.load()
    gen = torch.Generator().manual_seed(rows + Nout + K2)
    s = (torch.randn(rows, K1, generator=gen) * 2).to(DEV)
    s_in = torch.randn(rows, K2, generator=gen).to(DEV) if K2 else None
    K_ = K1 + K2
    W = (torch.randn(Nout, K_, generator=gen) / K_ ** 0.5).to(DEV)
    b = torch.randn(Nout, generator=gen).to(DEV)
    norm = (torch.rand(rows, generator=gen) + 0.1).to(DEV)
    gy = torch.randn(rows, Nout, generator=gen).to(DEV)
    base = torch.randn(rows, K1, generator=gen).to(DEV)
    res = {}
    try:
        assert lib.mrg_gemm_set_q(0) == 0           # the 16 x 16 x 32 kernel (round 5) sums k 32 at a time: equal to rounding, tested below
        for order in ((0, 0, 1), (0, 0, 0), (0, 2, 0), (2, 0, 1), (2, 1, 1)):
            assert lib.mrg_gemm_set_mode(order[0]) == 0 and lib.mrg_gemm_set_epilogue(order[1]) == 0 and lib.mrg_gemm_set_wide8(order[2]) == 0
            outs = []
            if K2 == 0:
                for act in (None, "relu", "sigmoid"):
                    outs.append(K.linear(s, W, b, act))
                # input gradient, plain and accumulating (gX += gY W)
                for acc in (0, 1):
                    gx = base.clone()
                    ws = torch.empty(max(16, int(lib.mrg_linear_bwd_input_workspace_bytes(K1, Nout))), dtype=torch.uint8, device=DEV)
                    call("mrg_linear_bwd_input", (ptr(gy), ptr(W), ptr(gx), ptr(ws), rows, K1, Nout, K1, acc, stream_of(gx)))
                    outs.append(gx)
            if Nout == K1:                                   # dense filters: gate (kind 0, stores the gate) and scale (kind 1)
                for kind in (0, 1):
                    out = torch.empty(rows, Nout, device=DEV)
                    gate = torch.empty(rows, Nout, device=DEV) if kind == 0 else None
                    ws = torch.empty(max(16, int(lib.mrg_gemm_workspace_bytes(K_, Nout))), dtype=torch.uint8, device=DEV)
                    call("mrg_dense_filter_fwd", (kind, ptr(s), ptr(s_in), ptr(W), ptr(b), ptr(norm), 1.0 / 3.0, ptr(out), ptr(gate), ptr(ws),
                                                  rows, K1, stream_of(out)))
                    outs += [out] + ([gate] if gate is not None else [])
            res[order] = outs
    finally:
        lib.mrg_gemm_set_epilogue(0)
        lib.mrg_gemm_set_mode(0)
        lib.mrg_gemm_set_wide8(1)
        lib.mrg_gemm_set_q(1)
    assert len(res[(0, 0, 1)]) > 0
    for other in ((0, 0, 0), (0, 2, 0), (2, 0, 1), (2, 1, 1)):
        assert len(res[other]) == len(res[(0, 0, 1)])
        for i, (x, y) in enumerate(zip(res[(0, 0, 1)], res[other])):
            assert torch.equal(x, y), (other, i, float((x - y).abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("rows,K1,K2,Nout", [(70001, 200, 0, 200), (66000, 200, 200, 200), (558771, 200, 0, 200), (33, 200, 0, 200), (64, 160, 0, 160),
                                             (100003, 400, 0, 200), (4097, 132, 132, 132), (20000, 64, 0, 224), (12345, 200, 0, 129), (1, 200, 0, 200)])
def test_three_waves_per_simd_row_gemm_agrees_with_the_two_wave_kernel(rows, K1, K2, Nout):
    """rowgemm_x3q_k (round 5: 16 x 16 x 32 tiles, 16 rows per wave, three workgroups per CU, weight half-slabs through a ring of two
    LDS buffers; the default for 129..224 output columns) against rowgemm_x3s_k (mrg_gemm_set_q(0)) and against float64: the same
    six cross terms, summed over k 32 at a time instead of 16, so the two agree to rounding of the f32 accumulation -- each output
    within 4e-6 of the output scale of the other, and the new kernel's error against float64 at most 1.5 x the old one's (+ 1e-6 of
    the scale).  Every epilogue the kernel has: bias + none / ReLU / sigmoid, accumulate, gate with and without the stored product,
    row scale; single and dual source, K % 32 != 0, ragged last strips, a single row, a partial last column pair, and the three
    direction segments in one grouped launch (mrg_dense_filter_fwd3 / mrg_linear_bwd_input3 / _pair)."""
    from mr_gnas_amd._lib import call, ptr, ptr_array, stream_of
    lib = mr_gnas_amd._lib.load()
    gen = torch.Generator().manual_seed(rows + Nout + K2 + 5)
    s = (torch.randn(rows, K1, generator=gen) * 2).to(DEV)
    s_in = torch.randn(rows, K2, generator=gen).to(DEV) if K2 else None
    K_ = K1 + K2
    W = (torch.randn(Nout, K_, generator=gen) / K_ ** 0.5).to(DEV)
    W3 = [(torch.randn(Nout, K_, generator=gen) / K_ ** 0.5).to(DEV) for _ in range(3)]
    b = torch.randn(Nout, generator=gen).to(DEV)
    b3 = [torch.randn(Nout, generator=gen).to(DEV) for _ in range(3)]
    norm = (torch.rand(rows, generator=gen) + 0.1).to(DEV)
    gy = torch.randn(rows, Nout, generator=gen).to(DEV)
    base = torch.randn(rows, K1, generator=gen).to(DEV)
    b0, b1 = rows // 3, rows - rows // 4
    x = s if s_in is None else torch.cat((s, s_in), 1)
    res, names = {}, []
    try:
        for q in (2, 0):                           # 2: every eligible K on the new kernel (the default, 1, takes K > 224 only)
            assert lib.mrg_gemm_set_q(q) == 0
            outs, names = [], []
            if K2 == 0:
                for act in (None, "relu", "sigmoid"):
                    outs.append(K.linear(s, W, b, act)); names.append(f"linear {act}")
                if Nout > 48:
                    for acc in (0, 1):
                        gx = base.clone()
                        ws = torch.empty(max(16, int(lib.mrg_linear_bwd_input_workspace_bytes(K1, Nout))), dtype=torch.uint8, device=DEV)
                        call("mrg_linear_bwd_input", (ptr(gy), ptr(W), ptr(gx), ptr(ws), rows, K1, Nout, K1, acc, stream_of(gx)))
                        outs.append(gx); names.append(f"bwd_input acc={acc}")
            if Nout == K1:
                for kind in (0, 1):
                    for store in ((True, False) if kind == 0 else (True,)):
                        out = torch.empty(rows, Nout, device=DEV) if store else None
                        gate = torch.empty(rows, Nout, device=DEV) if kind == 0 else None
                        ws = torch.empty(max(16, int(lib.mrg_gemm_workspace_bytes(K_, Nout))), dtype=torch.uint8, device=DEV)
                        if store:
                            call("mrg_dense_filter_fwd", (kind, ptr(s), ptr(s_in), ptr(W), ptr(b), ptr(norm), 1.0 / 3.0, ptr(out), ptr(gate), ptr(ws),
                                                          rows, K1, stream_of(s)))
                            outs += [out] + ([gate] if gate is not None else []); names += [f"filter kind={kind}"] + (["gate"] if gate is not None else [])
                        # the three direction segments in one grouped launch (gate only when store is False)
                        ws3n = int(lib.mrg_dense_filter3_workspace_bytes(Nout, K_))
                        if ws3n > 0:
                            out3 = torch.empty(rows, Nout, device=DEV) if store else None
                            gate3 = torch.empty(rows, Nout, device=DEV) if kind == 0 else None
                            ws3 = torch.empty(ws3n, dtype=torch.uint8, device=DEV)
                            call("mrg_dense_filter_fwd3", (kind, ptr(s), ptr(s_in), ptr_array(W3), ptr_array(b3), ptr(norm), 1.0 / 3.0, 1.0, ptr(out3), ptr(gate3),
                                                           ptr(ws3), b0, b1, rows, K1, stream_of(s)))
                            outs += ([out3] if store else []) + ([gate3] if gate3 is not None else [])
                            names += ([f"filter3 kind={kind}"] if store else []) + (["gate3"] if gate3 is not None else [])
            res[q] = outs
    finally:
        lib.mrg_gemm_set_q(1)
    assert len(res[2]) == len(res[0]) and len(res[2]) > 0
    for nm, a_, b_ in zip(names, res[2], res[0]):
        if a_.numel() == 0:
            continue
        scale = max(1.0, float(b_.abs().max()))
        assert float((a_ - b_).abs().max()) <= 4e-6 * scale, (nm, float((a_ - b_).abs().max()), scale)
    # against float64, for the plain product
    if K2 == 0 and rows > 0:
        ref = torch.nn.functional.linear(x.double(), W.double(), b.double())
        e_q, e_s = float((res[2][0].double() - ref).abs().max()), float((res[0][0].double() - ref).abs().max())
        assert e_q <= 1.5 * e_s + 1e-6 * float(ref.abs().max()), (e_q, e_s)


@pytest.mark.gpu
@pytest.mark.parametrize("D", [200, 256])
@pytest.mark.parametrize("kind", ["max", "mean"])
def test_fused_aggregators_are_bit_exact_across_split_core_kernels(kind, D):
    """Fused a_max / a_mean (GEMM over gathered, destination-ordered edge rows with the segmented epilogue) on the default
    LDS-weight kernel (gathered rows read straight into registers) against the wave-autonomous kernel (mode 2, gathered rows by
    LDS-DMA): outputs and every gradient bit-identical.  D = 256 runs the eight-tile column block (gemm_x3s8.hpp) in mode 0."""
    lib = mr_gnas_amd._lib.load()
    N, E, R = 2500, 120000, 9
    gen = torch.Generator().manual_seed(77)
    src = torch.randint(0, N, (E,), generator=gen)
    dst = torch.randint(0, N - 10, (E,), generator=gen)
    dst[: E // 3] = 5                                               # a hub that spans many strips
    et = torch.randint(0, R, (E,), generator=gen)
    g = G.RelGraph(N, src.numpy(), dst.numpy(), et.numpy(), np.ones(E, np.float32), device=DEV)
    x0 = torch.randn(E + N, D, generator=gen)
    W0 = torch.randn(D, D, generator=gen) / D ** 0.5
    b0 = torch.randn(D, generator=gen) * 0.1
    gout = torch.randn(N, D, generator=gen).to(DEV)
    res = {}
    try:
        for mode in (0, 2):
            assert lib.mrg_gemm_set_mode(mode) == 0
            x, W, b = (t.clone().to(DEV).requires_grad_(True) for t in (x0, W0, b0))
            out = K.linear_relu_aggregate(kind, x, W, b, g)
            out.backward(gout)
            res[mode] = (out.detach(), x.grad, W.grad, b.grad)
    finally:
        lib.mrg_gemm_set_mode(0)
    for a, b_, what in zip(res[0], res[2], ("out", "gx", "gW", "gb")):
        assert torch.equal(a, b_), what


@pytest.mark.gpu
@pytest.mark.parametrize("N,E,R,D,train", [(3000, 90000, 11, 200, True), (400, 5000, 5, 64, True), (300, 3000, 4, 10, True), (500, 20000, 7, 200, False)])
def test_cell_zero_recompute_matches_the_stored_candidates(N, E, R, D, train):
    """Cell_Zero's MixedOp over PRE_OPS with the candidates recomputed from the tables (mrg_zero_*: no [rows, D] candidate is
    stored) against three gather-compose launches + the generic epilogue: output, BatchNorm running statistics, alpha / gamma /
    beta gradients bit-identical (same values, same summation order); the table gradients (the association of their sums
    differs) within rounding, and against the plain torch formulation."""
    from mr_gnas_amd import supernet as S
    gen = torch.Generator().manual_seed(N + E + D)
    ent0 = torch.randn(N, D, generator=gen)
    rel0 = torch.randn(2 * R + 1, D, generator=gen)
    ei = torch.randint(0, N, (E + N,), generator=gen).to(DEV)
    ri = torch.randint(0, 2 * R + 1, (E + N,), generator=gen).to(DEV)
    w0 = torch.softmax(torch.randn(3, generator=gen), 0)
    gout = torch.randn(E + N, D, generator=gen).to(DEV)
    res = {}
    try:
        for fused in (True, False):
            K.switches.CELL_ZERO_FUSED = fused
            torch.manual_seed(5)
            mixed = S.MixedOp(D, 0.0, O.PRE_OPS).to(DEV)
            with torch.no_grad():
                for _, bn, _ in mixed._ops:
                    bn.weight.add_(0.1 * torch.randn_like(bn.weight)); bn.bias.add_(0.1 * torch.randn_like(bn.bias))
                    bn.running_mean.add_(0.05); bn.running_var.mul_(1.3)
            mixed.train(train)
            ent, rel, w = (t.clone().to(DEV).requires_grad_(True) for t in (ent0, rel0, w0))
            gp_e, gp_r = K.GatherPlan(ei, N), K.GatherPlan(ri, 2 * R + 1)
            out = mixed(w, None, K.LazyRows(ent, gp_e), K.LazyRows(rel, gp_r))
            out.backward(gout)
            res[fused] = dict(out=out.detach(), w=w.grad, ent=ent.grad, rel=rel.grad,
                              gam=[bn.weight.grad.clone() for _, bn, _ in mixed._ops], bet=[bn.bias.grad.clone() for _, bn, _ in mixed._ops],
                              rm=[bn.running_mean.clone() for _, bn, _ in mixed._ops], rv=[bn.running_var.clone() for _, bn, _ in mixed._ops])
            if fused:
                model = mixed
    finally:
        K.switches.CELL_ZERO_FUSED = True
    a, b = res[True], res[False]
    # same values; the recomputing statistics / gradient reductions sum over more blocks than the stored form's (order of the
    # float64 / float32 partial sums): equal to rounding, not bit for bit
    close(a["out"], b["out"].cpu(), "cell zero output", rtol=2e-6, atol=2e-6)
    close(a["w"], b["w"].cpu(), "cell zero d alpha", rtol=2e-5, atol=2e-5 * float(b["w"].abs().max()))
    for key in ("gam", "bet", "rm", "rv"):
        for x, y in zip(a[key], b[key]):
            close(x, y.cpu(), "cell zero " + key, rtol=2e-5, atol=2e-5 * max(1e-3, float(y.abs().max())))
    for key in ("ent", "rel"):
        close(a[key], b[key].cpu(), "cell zero table gradient " + key, rtol=2e-5, atol=2e-5 * float(b[key].abs().max()))
    # the torch formulation (reference models/cell_lp.py:25-33 on pre_mult / pre_sub / pre_add of the gathered rows)
    e64, r64, w64 = (t.double().to(DEV).requires_grad_(True) for t in (ent0, rel0, w0))
    x, h = e64[ei.long()], r64[ri.long()]
    ref = 0
    for k, (y, (_, bn, _)) in enumerate(zip((x * h, x - h, x + h), model._ops)):
        if train:
            mu, var = y.mean(0), y.var(0, unbiased=False)
        else:
            mu, var = (bn.running_mean.double() - 0.0), bn.running_var.double()
        if train:
            z = (y - mu) / torch.sqrt(var + bn.eps) * bn.weight.double() + bn.bias.double()
        else:
            z = None
        if z is not None:
            ref = ref + w64[k] * torch.relu(z)
    if train:
        ref.backward(gout.double())
        close(a["out"], ref.detach().float().cpu(), "cell zero vs torch float64", rtol=1e-4, atol=1e-5)
        close(a["ent"], e64.grad.float().cpu(), "cell zero d ent vs torch float64", rtol=1e-4, atol=1e-4 * float(e64.grad.abs().max()))
        close(a["rel"], r64.grad.float().cpu(), "cell zero d rel vs torch float64", rtol=1e-4, atol=1e-4 * float(r64.grad.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("tied", [True, False])
@pytest.mark.parametrize("N,E,R,D", [(2000, 150000, 9, 200), (300, 5000, 4, 64)])
def test_dense_pair_node_matches_the_two_operators(N, E, R, D, tied):
    """A first-stage MixedOp with f_dense_comp and f_comp as ONE autograd node (one input-gradient product over the
    concatenated reduction dimension, mrg_linear_bwd_input3_pair) against the two operators as separate nodes: forward
    bit-identical, every gradient within float32 summation-order tolerance; candidates on side streams (large case) and
    on one stream (small case); tied operands (h is h_in) and distinct ones."""
    from mr_gnas_amd import supernet as S
    gen = torch.Generator().manual_seed(N + E + D + int(tied))
    src = torch.randint(0, N, (E,), generator=gen)
    dst = torch.randint(0, N, (E,), generator=gen)
    et = torch.randint(0, 2 * R, (E,), generator=gen)
    g = G.RelGraph(N, src.numpy(), dst.numpy(), et.numpy(), (torch.rand(E, generator=gen) + 0.1).numpy().astype(np.float32), device=DEV)
    h0 = torch.randn(E + N, D, generator=gen)
    hin0 = h0 if tied else torch.randn(E + N, D, generator=gen)
    w0 = torch.softmax(torch.randn(len(O.FIRST_OPS), generator=gen), 0)
    gout = torch.randn(E + N, D, generator=gen).to(DEV)
    torch.manual_seed(3)
    mixed = S.MixedOp(D, 0.0, O.FIRST_OPS).to(DEV)
    S.xavier_init_(mixed)
    res = {}
    try:
        for pair in (True, False, "no identity fold", "neither"):
            K.switches.DENSE_PAIR = pair in (True, "no identity fold")
            K.switches.FOLD_IDENTITY = pair in (True, False)          # f_identity's gradient added into f_dense_comp's direct term / a tensor of its own
            mixed.zero_grad(set_to_none=True)
            h = h0.clone().to(DEV).requires_grad_(True)
            hin = h if tied else hin0.clone().to(DEV).requires_grad_(True)
            w = w0.clone().to(DEV).requires_grad_(True)
            out = mixed(w, g, h, hin)
            out.backward(gout)
            torch.cuda.synchronize()
            res[pair] = [out.detach(), h.grad] + ([] if tied else [hin.grad]) + [w.grad] + [p.grad.clone() for p in mixed.parameters()]
    finally:
        K.switches.DENSE_PAIR, K.switches.FOLD_IDENTITY = True, True
    for other in (False, "no identity fold", "neither"):
        assert torch.equal(res[True][0], res[other][0])
        for i, (a, b) in enumerate(zip(res[True][1:], res[other][1:])):
            close(a, b.cpu(), f"dense pair gradient {i} vs {other}", rtol=3e-5, atol=3e-5 * max(1.0, float(b.abs().max())))


@pytest.mark.gpu
@pytest.mark.parametrize("D", [64, 200, 512])
def test_row_factor_candidate_without_a_gated_partner_is_multiplied_out(D):
    """f_sparse_comp as a row factor handed to an epilogue that cannot recompute it (no gate-only f_dense_comp of the same rows; D
    beyond one lane group per row): mixed_epilogue multiplies s * fvec out itself -- same output bit for bit as the stored
    operator, every gradient within float32 rounding (the node then forms dz (x) u for the gradient w.r.t. s itself)."""
    N, E, R = 120, 2500, 4
    gen = torch.Generator().manual_seed(D)
    g = G.RelGraph(N, torch.randint(0, N, (E,), generator=gen).numpy(), torch.randint(0, N, (E,), generator=gen).numpy(),
                   torch.randint(0, 2 * R, (E,), generator=gen).numpy(), (torch.rand(E, generator=gen) + 0.1).numpy().astype(np.float32), device=DEV)
    op = O.MIXED_OPS["f_sparse_comp"]({"feature_dim": D}).to(DEV)
    for p in op.parameters():
        p.data.add_(0.05 * torch.randn(p.shape, generator=gen).to(DEV))
    bn = torch.nn.BatchNorm1d(D).to(DEV)
    x0, xin0 = torch.randn(E + N, D, generator=gen), torch.randn(E + N, D, generator=gen)
    gout = torch.randn(E + N, D, generator=gen).to(DEV)
    res = {}
    for row in (True, False):
        op.zero_grad(set_to_none=True)
        bn.zero_grad(set_to_none=True)
        bn.reset_running_stats()
        x, xin = x0.clone().to(DEV).requires_grad_(True), xin0.clone().to(DEV).requires_grad_(True)
        w = torch.ones(1, device=DEV, requires_grad=True)
        y = op(g, x, xin, for_epilogue=row)
        assert isinstance(y, K.Candidate) == row and (not row or (y.kind == "rowfactor" and y.y.dim() == 1))
        out = K.mixed_epilogue([y], [bn], w, fold_row_scales=True)
        out.backward(gout)
        torch.cuda.synchronize()
        res[row] = [out.detach(), x.grad, xin.grad, w.grad, bn.weight.grad, bn.bias.grad] + [p.grad.clone() for p in op.parameters()]
    assert torch.equal(res[True][0], res[False][0])
    for i, (a, b) in enumerate(zip(res[True][1:], res[False][1:])):
        close(a, b.cpu(), f"multiplied-out row factor: gradient {i}", rtol=2e-5, atol=2e-5 * max(1e-3, float(b.abs().max())))


@pytest.mark.gpu
@pytest.mark.parametrize("tied", [True, False])
@pytest.mark.parametrize("N,E,R,D", [(2000, 150000, 9, 200), (300, 5000, 4, 64), (50, 700, 3, 100)])
def test_gate_only_candidate_is_bit_exact_with_the_stored_one(N, E, R, D, tied):
    """f_dense_comp's output never stored (the row GEMM writes only the gate, the MixedOp epilogue's four passes recompute
    gate * s * c: mrg_gated_branch) against the stored candidate: output, running statistics and EVERY gradient bit-identical --
    with the epilogue's gradient folds on and off, in training and in eval mode.  On top of it f_sparse_comp as a row factor
    (y = s * fvec[r] recomputed, its gradient w.r.t. s added by the epilogue's gradient store: mrg_gated_branch.row_k): output and
    running statistics bit-identical, gradients within float32 rounding of the stored form (the association of dz differs)."""
    from mr_gnas_amd import supernet as S
    gen = torch.Generator().manual_seed(7 * N + E + D + int(tied))
    src = torch.randint(0, N, (E,), generator=gen)
    dst = torch.randint(0, N, (E,), generator=gen)
    et = torch.randint(0, 2 * R, (E,), generator=gen)
    g = G.RelGraph(N, src.numpy(), dst.numpy(), et.numpy(), (torch.rand(E, generator=gen) + 0.1).numpy().astype(np.float32), device=DEV)
    h0 = torch.randn(E + N, D, generator=gen)
    hin0 = h0 if tied else torch.randn(E + N, D, generator=gen)
    w0 = torch.softmax(torch.randn(len(O.FIRST_OPS), generator=gen), 0)
    gout = torch.randn(E + N, D, generator=gen).to(DEV)
    add0 = torch.randn(E + N, D, generator=gen).to(DEV)
    torch.manual_seed(5)
    mixed = S.MixedOp(D, 0.0, O.FIRST_OPS).to(DEV)
    S.xavier_init_(mixed)
    for p in mixed.parameters():                          # biases and gate vectors away from their all-zero / symmetric start
        if p.dim() == 1:
            p.data.add_(0.1 * torch.randn(p.shape, generator=gen).to(DEV))
    state0 = {k: v.clone() for k, v in mixed.state_dict().items()}
    calls = []
    import types
    real_call = mr_gnas_amd._lib.call
    launchers = [m for m in vars(K).values() if isinstance(m, types.ModuleType) and getattr(m, "call", None) is real_call]

    def set_call(fn):                                     # every kernel-family module of the package binds `call` by name
        for m in launchers:
            m.call = fn

    def spy(name, args, **kw):
        calls.append((name, args))
        return real_call(name, args, **kw)

    try:
        for folds in (True, False):
            for training in (True, False):
                res = {}
                for mode in ("stored", "gate", "gate+row"):
                    K.switches.GATED_RECOMPUTE, K.switches.ROW_FACTOR = mode != "stored", mode == "gate+row"
                    K.switches.FOLD_ROW_SCALE, K.switches.FOLD_IDENTITY = folds, folds
                    mixed.load_state_dict(state0)
                    mixed.train(training)
                    mixed.zero_grad(set_to_none=True)
                    h = h0.clone().to(DEV).requires_grad_(True)
                    hin = h if tied else hin0.clone().to(DEV).requires_grad_(True)
                    w = w0.clone().to(DEV).requires_grad_(True)
                    del calls[:]
                    set_call(spy)
                    out = mixed(w, g, h, hin, addend=add0)
                    out.backward(gout)
                    set_call(real_call)
                    torch.cuda.synchronize()
                    fwd3 = [a for n, a in calls if n == "mrg_dense_filter_fwd3" and a[0] == 0]
                    assert len(fwd3) == 1 and (fwd3[0][8] is None) == (mode != "stored"), "the gated row GEMM stores its output exactly when the candidate is stored"
                    names = [n for n, _ in calls]
                    rowed = mode == "gate+row" and folds          # the row factor needs the gated candidate's folded gradient store
                    assert ("mrg_gate_row_fwd" in names) == rowed and ("mrg_gate_row_bwd" in names) == rowed
                    assert ("mrg_gate_fwd" in names) == (not rowed)
                    res[mode] = ([out.detach(), h.grad] + ([] if tied else [hin.grad]) + [w.grad] + [p.grad.clone() for p in mixed.parameters()]
                                 + [b.clone() for b in mixed.buffers()])
                for i, (a, b) in enumerate(zip(res["gate"], res["stored"])):
                    assert torch.equal(a, b), f"gate-only vs stored f_dense_comp: tensor {i} differs (folds {folds}, training {training})"
                nb = len(list(mixed.buffers()))
                assert torch.equal(res["gate+row"][0], res["stored"][0])
                for a, b in zip(res["gate+row"][-nb:], res["stored"][-nb:]):
                    assert torch.equal(a, b), "running statistics"
                for i, (a, b) in enumerate(zip(res["gate+row"][1:-nb], res["stored"][1:-nb])):
                    close(a, b.cpu(), f"row-factor f_sparse_comp: gradient {i} (folds {folds}, training {training})", rtol=2e-5,
                          atol=2e-5 * max(1e-3, float(b.abs().max())))
    finally:
        set_call(real_call)
        K.switches.GATED_RECOMPUTE, K.switches.ROW_FACTOR, K.switches.FOLD_ROW_SCALE, K.switches.FOLD_IDENTITY = True, True, True, True


@pytest.mark.parametrize("max_norm,wd", [(5.0, 0.0), (0.05, 0.0), (0.0, 0.0), (1.0, 3e-4)])
def test_clipped_sgd_equals_torch_clip_grad_norm_and_sgd(max_norm, wd):
    """optim.ClippedSGD (mrg_clip_sgd_step: three launches) against torch.nn.utils.clip_grad_norm_ + torch.optim.SGD(momentum)
    (reference search/mr_lp_search.py:118-119,243-245) over four steps: ragged tensor sizes around the chunk size, a parameter that
    never receives a gradient, one that misses it in one step, clipping active / inactive / off, weight decay."""
    from mr_gnas_amd.optim import ClippedSGD
    gen = torch.Generator(device=DEV).manual_seed(3)
    shapes = [(200, 400), (200,), (1,), (4096,), (4097,), (3, 5, 7), (14541, 100), (475, 200), (8191,)]
    mine = [torch.randn(*s, device=DEV, generator=gen).requires_grad_(True) for s in shapes]
    ref = [p.detach().clone().requires_grad_(True) for p in mine]
    opt = ClippedSGD(mine, 1e-2, momentum=0.9, weight_decay=wd, max_norm=max_norm)
    topt = torch.optim.SGD(ref, 1e-2, momentum=0.9, weight_decay=wd)
    for step in range(4):
        for i, (a, b) in enumerate(zip(mine, ref)):
            if i == 2 or (i == 4 and step == 1):
                a.grad = b.grad = None
                continue
            g = torch.randn(a.shape, device=DEV, generator=gen) * (0.01 if step == 3 else 1.0)
            a.grad, b.grad = g.clone(), g.clone()
        if max_norm > 0:
            tn = torch.nn.utils.clip_grad_norm_(ref, max_norm)
        else:
            tn = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(p.grad) for p in ref if p.grad is not None]))
        topt.step()
        nc = opt.step()
        assert abs(float(nc[0]) - float(tn)) <= 1e-5 * float(tn)
        for i, (a, b) in enumerate(zip(mine, ref)):
            close(a.detach(), b.detach().cpu(), f"step {step} parameter {i}", rtol=2e-6, atol=2e-7, rms_rtol=2e-6)
    assert torch.equal(mine[2], ref[2])                      # never had a gradient: untouched, bit for bit
    with pytest.raises(Exception):
        ClippedSGD([torch.zeros(3)], 0.1)


@pytest.mark.parametrize("D", [200, 100, 160])
def test_few_rows_row_gemm_is_bit_exact(D):
    """Products of at most 4 096 rows run on the wave-autonomous kernel with two-tile column blocks (gemm_dispatch.hpp: x3n_shape,
    mrg_gemm_set_small): every operator that owns a row GEMM -- plain, fused a_max / a_mean (segmented epilogues, gathered rows),
    the three-segment dense filters tied and untied, their input and weight gradients -- must give the SAME BITS as with the
    switch off (one kernel for every row count)."""
    lib = mr_gnas_amd._lib.load()
    gen = torch.Generator().manual_seed(D)
    N, E, R = 300, 1100, 5
    src, dst = torch.randint(0, N, (E,), generator=gen), torch.randint(0, N, (E,), generator=gen)
    dst[:200] = 7                                             # a hub
    et = torch.randint(0, 2 * R, (E,), generator=gen)
    g = G.RelGraph(N, src.numpy(), dst.numpy(), et.numpy(), (torch.rand(E, generator=gen) + 0.1).numpy().astype(np.float32), device=DEV)
    a0, b0 = torch.randn(E + N, D, generator=gen), torch.randn(E + N, D, generator=gen)
    gM, gN = torch.randn(E + N, D, generator=gen).to(DEV), torch.randn(N, D, generator=gen).to(DEV)
    res = {}
    try:
        for small in (1, 0):
            assert lib.mrg_gemm_set_small(small) == 0
            torch.manual_seed(1)
            outs = []
            for kind, tied in (("f_dense_comp", True), ("f_dense_comp", False), ("f_comp", True), ("a_max", False), ("a_mean", False),
                               ("f_dense_last", False)):
                torch.manual_seed(hash(kind) % 1000)
                op = O.MIXED_OPS[kind]({"feature_dim": D, "drop_aggr": 0.0}).to(DEV)
                node = kind == "f_dense_last"
                a = (a0[:N] if node else a0).clone().to(DEV).requires_grad_(True)
                b = a if tied else (b0[:N] if node else b0).clone().to(DEV).requires_grad_(True)
                out = op(g, a, b)
                out.backward(gN if out.shape[0] == N else gM)
                outs += [out.detach(), a.grad] + [p.grad.clone() for p in op.parameters() if p.grad is not None]
            x = a0[:900].clone().to(DEV).requires_grad_(True)
            W = torch.randn(D, D, generator=gen).to(DEV).requires_grad_(True) if small else W
            y = K.linear(x, W, None, "relu")
            y.backward(gM[:900])
            outs += [y.detach(), x.grad, W.grad.clone()]
            W.grad = None
            res[small] = outs
    finally:
        lib.mrg_gemm_set_small(1)
    assert len(res[1]) == len(res[0]) and len(res[1]) > 20
    for i, (p, q) in enumerate(zip(res[1], res[0])):
        assert torch.equal(p, q), f"tensor {i}: max diff {float((p - q).abs().max()):.3e}"


def test_clipped_sgd_replays_from_a_hip_graph():
    """backward + ClippedSGD.step() captured once and replayed: the gradient pointer table is copied from a pinned buffer of the
    capture's own (no pinning while a capture is open), the gradients live at fixed addresses of the graph's pool."""
    from mr_gnas_amd.optim import ClippedSGD
    gen = torch.Generator(device=DEV).manual_seed(5)
    W = [torch.randn(64, 200, device=DEV, generator=gen).requires_grad_(True), torch.randn(5000, device=DEV, generator=gen).requires_grad_(True)]
    ref = [w.detach().clone().requires_grad_(True) for w in W]
    x = torch.randn(32, 200, device=DEV, generator=gen)
    opt = ClippedSGD(W, 1e-2, momentum=0.9, max_norm=1.0)
    topt = torch.optim.SGD(ref, 1e-2, momentum=0.9)

    def step(ws, o, clip):
        loss = (x @ ws[0].t()).square().mean() + ws[1].square().sum() * 1e-3
        loss.backward()
        if clip:
            torch.nn.utils.clip_grad_norm_(ws, 1.0)
        o.step()
        o.zero_grad(set_to_none=True)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step(W, opt, False)                                   # warm-up (eager)
    torch.cuda.current_stream().wait_stream(side)
    step(ref, topt, True)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        step(W, opt, False)
    for _ in range(3):
        graph.replay()
        step(ref, topt, True)
    torch.cuda.synchronize()
    for a, b in zip(W, ref):
        close(a.detach(), b.detach().cpu(), "parameter after 1 eager + 3 replayed steps", rtol=5e-6, atol=1e-6, rms_rtol=5e-6)


@pytest.mark.parametrize("seed", range(12))
def test_segmax_bwd_input_random_shapes(seed):
    """mrg_segmax_bwd_input straight through the C ABI on random shapes (D and K multiples of 4 up to the LDS bound, K != D, tiny and
    empty edge lists, hubs, isolated nodes, a maximum that is not positive, destination-ordered and edge-id walks, no ReLU mask):
    gmsg bit-exact with its definition, gx against the float64 product of that gmsg with W."""
    from mr_gnas_amd._lib import call, ptr, stream_of
    lib = mr_gnas_amd._lib.load()
    rng = np.random.default_rng(100 + seed)
    D = int(rng.choice([4, 8, 52, 64, 100, 128, 200]))
    Kin = int(rng.choice([4, 16, 64, 100, 200])) if D * 200 * 4 <= 160 * 1024 else int(rng.choice([4, 64]))
    if D * Kin * 4 > 160 * 1024:
        Kin = 4
    N = int(rng.choice([1, 7, 300, 5000]))
    E = int(rng.choice([0, 1, 15, 17, 1000, 60000]))
    assert lib.mrg_segmax_bwd_input_ok(D, Kin) == 1
    gen = torch.Generator(device=DEV).manual_seed(seed)
    dst = torch.randint(0, N, (max(E, 1),), device=DEV, generator=gen, dtype=torch.int32)[:E]
    if E > 20 and N > 3:
        dst[: E // 3] = 2                                       # a hub
    # a consistent arg table: for every (node, column) one of the node's in-edges (or -1 without in-edges)
    arg = torch.full((N, D), -1, dtype=torch.int32, device=DEV)
    if E:
        pick = torch.rand(E, D, device=DEV, generator=gen)
        best = torch.full((N, D), -1.0, device=DEV).scatter_reduce(0, dst.long().view(-1, 1).expand(E, D), pick, "amax", include_self=True)
        win = pick == best[dst.long()]
        eid = torch.arange(E, device=DEV, dtype=torch.int32).view(-1, 1).expand(E, D)
        arg = torch.full((N, D), 2**31 - 1, dtype=torch.int32, device=DEV).scatter_reduce(0, dst.long().view(-1, 1).expand(E, D),
                                                                                      torch.where(win, eid, 2**31 - 1), "amin")
        arg = torch.where(arg == 2**31 - 1, -1, arg).contiguous()
    gout = torch.randn(N, D, device=DEV, generator=gen)
    mx = torch.randn(N, D, device=DEV, generator=gen) if seed % 3 else None          # a third of the cases: no ReLU mask
    W = torch.randn(D, Kin, device=DEV, generator=gen)
    order = torch.argsort(dst.long(), stable=True).int() if (E and seed % 2) else None
    gmsg = torch.full((max(E, 1), D), float("nan"), device=DEV)[:E]
    gx = torch.full((max(E, 1), Kin), float("nan"), device=DEV)[:E]
    call("mrg_segmax_bwd_input", (ptr(gout), ptr(mx), ptr(dst), ptr(arg), ptr(W), ptr(gmsg), ptr(gx), ptr(order), E, N, D, Kin, stream_of(gout)))
    torch.cuda.synchronize()
    if E == 0:
        return
    eids = torch.arange(E, device=DEV, dtype=torch.int32).view(-1, 1)
    keep = (arg[dst.long()] == eids) & ((mx[dst.long()] > 0) if mx is not None else True)
    ref_msg = torch.where(keep, gout[dst.long()], 0.0)
    assert torch.equal(gmsg, ref_msg)
    ref_gx = (ref_msg.double() @ W.double())
    err = float((gx.double() - ref_gx).abs().max())
    assert err <= 2e-6 * max(float(ref_gx.abs().max()), 1.0) * max(1.0, (keep.sum(1).max().item()) ** 0.5), (err, D, Kin, N, E)
    # gmsg == NULL: only gx
    gx2 = torch.empty_like(gx)
    call("mrg_segmax_bwd_input", (ptr(gout), ptr(mx), ptr(dst), ptr(arg), ptr(W), None, ptr(gx2), ptr(order), E, N, D, Kin, stream_of(gout)))
    assert torch.equal(gx2, gx)
