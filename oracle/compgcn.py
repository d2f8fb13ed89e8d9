"""Oracle CompGCN layer / stack: functional restatement of reference
models/compgcn.py:12-185 (CPU, test-only; see oracle/__init__.py)."""
import torch
import torch.nn.functional as F

from .graph import seg_sum


def ccorr(a, b):
    """Circular correlation out[k] = sum_i a[i] * b[(i+k) % D].

    Reference utils/utils.py:274-301 (== models/operations_lp.py:47-59) writes
    it as irfft(conj(rfft(a)) * rfft(b)) with the pre-1.8 torch FFT API; this is
    the same formula on the current API."""
    D = a.shape[-1]
    return torch.fft.irfft(torch.conj(torch.fft.rfft(a, dim=-1)) * torch.fft.rfft(b, dim=-1), n=D, dim=-1)


def ccorr_direct(a, b):
    """The definition itself, float64 accumulate (what the fixtures hold)."""
    D = a.shape[-1]
    idx = (torch.arange(D).view(-1, 1) + torch.arange(D).view(1, -1)) % D
    bb = torch.broadcast_to(b, torch.broadcast_shapes(a.shape, b.shape)).double()
    return torch.einsum("...i,...ik->...k", a.double().expand_as(bb), bb[..., idx]).float()


def compose(kind, u, e, corr=ccorr):
    # reference models/compgcn.py:62-69 and :90-97
    if kind == "sub":
        return u - e
    if kind == "mul":
        return u * e
    if kind == "ccorr":
        return corr(u, e.expand_as(u) if e.dim() < u.dim() else e)
    raise Exception("Only supports sub, mul, and ccorr")


def comp_graph_conv(g, P, n_in, r_in, in_mask, comp_fn="sub", batchnorm=True, dropout=0.0,
                    training=True, prefix="", corr=ccorr):
    """One CompGCN layer, reference models/compgcn.py:48-113.

    g.etype indexes ``cat(r_in, loop_rel)``; ``in_mask`` (bool [E]) selects the
    edges transformed by W_I, its complement those transformed by W_O.
    Returns (n_out [N, Dout], r_out [R', Dout] without the self-loop row)."""
    p = lambda k: P[prefix + k]
    r = torch.cat((r_in, p("loop_rel")), 0)                                    # :55
    e_feat = r[g.etype] * g.norm.view(-1, 1)                                   # :58
    comp = compose(comp_fn, n_in[g.src], e_feat, corr)                         # :62-69
    in_idx = torch.nonzero(in_mask, as_tuple=False).squeeze(-1)                # :74
    out_idx = torch.nonzero(~in_mask, as_tuple=False).squeeze(-1)              # :75
    new = torch.zeros(comp.shape[0], p("W_O.weight").shape[0], dtype=comp.dtype, device=comp.device)
    new = new.index_put((out_idx,), F.linear(comp[out_idx], p("W_O.weight"), p("W_O.bias")))   # :77,81
    new = new.index_put((in_idx,), F.linear(comp[in_idx], p("W_I.weight"), p("W_I.bias")))     # :78,82
    comp_edge = seg_sum(new, g.dst, g.n)                                        # :87
    comp_s = compose(comp_fn, n_in, r[-1], corr)                                # :90-97
    n_out = (F.linear(comp_s, p("W_S.weight"), p("W_S.bias")) + F.dropout(comp_edge, dropout, training)) * (1 / 3)
    r_out = F.linear(r, p("W_R.weight"), p("W_R.bias"))                         # :103
    if batchnorm:                                                               # :106-107
        n_out = F.batch_norm(n_out, None, None, p("bn.weight"), p("bn.bias"), training=True)
    return torch.tanh(n_out), r_out[:-1]                                        # :110-113


def comp_gcn(g, P, in_mask, num_layers, comp_fn="sub", batchnorm=True, dropout=0.0,
             layer_dropout=None, training=True, corr=ccorr):
    """Reference models/compgcn.py:172-185."""
    n = P["n_embds"]
    r = torch.mm(P["weights"], P["basis"]) if "basis" in P else P["rel_embds"]   # :175-179
    for i in range(num_layers):
        n, r = comp_graph_conv(g, P, n, r, in_mask, comp_fn, batchnorm, dropout, training,
                               prefix=f"layers.{i}.", corr=corr)
        n = F.dropout(n, (layer_dropout or [0.0] * num_layers)[i], training)
    return n, r
