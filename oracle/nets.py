"""Oracle callers of the hot path: the fixed-genotype network/cell (reference
models/model_lp.py) and the mixed-op supernet (reference models/cell_lp.py,
models/model_search_lp.py), restated functionally over a ``{state_dict key:
tensor}`` dict ``S`` (CPU, test-only; see oracle/__init__.py).

All BatchNorms run in training mode (batch statistics), dropout p = 0 unless
given -- this is the configuration of the golden vectors and of the bench.
"""
import torch
import torch.nn.functional as F

from . import ops as O


# Test instrumentation ("mask replay", tests/test_configs_gpu.py): RELU_HOOK(site, z) -> tensor replaces F.relu(z) at the ReLU
# sites of the supernet -- site = the BatchNorm's state_dict prefix ("cells.0.cell_first._ops.1._ops.2.1.") or ("net", layer) --
# so that a float64 run can be made to take another run's ReLU decisions.  None = plain F.relu.
RELU_HOOK = None


def _relu(site, z):
    return F.relu(z) if RELU_HOOK is None else RELU_HOOK(site, z)


def _sub(S, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in S.items() if k.startswith(prefix)}


def _bn(S, prefix, x):
    return F.batch_norm(x, None, None, S[prefix + "weight"], S[prefix + "bias"], training=True)


def _embed_tables(S):
    # reference models/model_lp.py:124-125 == models/model_search_lp.py:133-134
    ent = F.linear(S["embedding_h.weight"], S["linear_e.weight"], S["linear_e.bias"])
    rel = torch.mm(S["rel_wt"], S["embedding_e.weight"])
    return ent, rel


# ---------------------------------------------------------------------------
# fixed genotype (config 1)
# ---------------------------------------------------------------------------
def fixed_cell(g, S, prefix, genotype, x, hr):
    """Reference models/model_lp.py:59-74 with OpModule.forward :27-35."""
    edges = {}
    for name, center, pre in genotype.alpha_cell:
        edges[(center - 1, pre)] = name
    nb = len(set(c for _, c, _ in genotype.alpha_cell))

    def run(center, pre, a, b):
        name = edges[(center, pre)]
        pp = f"{prefix}_ops.{center}.{pre}.0."
        h = O.OPS[name](g, _sub(S, pp + "op."), a, b)
        if name != "pre_mult":               # model_lp.py:31 -- only pre_mult skips BN+ReLU
            h = F.relu(_bn(S, pp + "batchnorm_h.", h))
        return h

    zero_out = run(0, 0, x, hr)
    states = [x, zero_out]
    for n in range(1, nb):
        hs = [run(n, i, states[i], zero_out) for i in range(n + 1) if (n, i) in edges]
        states.append(sum(hs))
    concat = genotype.concat_node if genotype.concat_node is not None else list(range(1, 1 + nb))
    h = F.linear(torch.cat([states[i] for i in concat], dim=1), S[prefix + "concat.weight"], S[prefix + "concat.bias"])
    return F.relu(_bn(S, prefix + "batchnorm_h.", h))


def fixed_net_forward(g, S, genotypes, subj, rel, num_rel_rows, gamma=40.0):
    """Reference models/model_lp.py:123-137. Returns the score matrix [B, N]."""
    ent, rel_emb = _embed_tables(S)
    dev = g.src.device
    src_f = torch.cat((g.src, torch.arange(g.n, device=dev)))
    et_f = torch.cat((g.etype, torch.full((g.n,), num_rel_rows - 1, dtype=torch.long, device=dev)))
    for i, geno in enumerate(genotypes):
        ent = fixed_cell(g, S, f"cells.{i}.", geno, ent[src_f], rel_emb[et_f])
        rel_emb = torch.matmul(rel_emb, S["w_rel"])
    return O.SF[genotypes[-1].score_func](ent, ent[subj], rel_emb[rel], gamma)


# ---------------------------------------------------------------------------
# supernet (config 2)
# ---------------------------------------------------------------------------
def mixed_op(g, S, prefix, names, w, a, b):
    """Reference models/cell_lp.py:25-33: sum_k w_k * ReLU(BN_k(op_k(g, a, b)))."""
    out = 0
    for k, name in enumerate(names):
        h = O.OPS[name](g, _sub(S, f"{prefix}_ops.{k}.0."), a, b)
        h = h if h.dtype == torch.float64 else h.float()      # reference: `.float()`; kept float64 when the tests run the checker in float64
        out = out + w[k] * _relu(f"{prefix}_ops.{k}.1.", _bn(S, f"{prefix}_ops.{k}.1.", h))
    return out


def super_cell(g, S, prefix, nfirst, nlast, x, hr, Wz, Wf, Wm, Wl):
    """Reference models/cell_lp.py:173-188 with Cell_Zero/First/Middle/Last :53-152."""
    h_in = mixed_op(g, S, prefix + "cell_zero._ops.0.", O.PRE_OPS, Wz[0], x, hr)
    states, off = [h_in], 0
    for _ in range(nfirst):                                           # Cell_First
        s = sum(mixed_op(g, S, f"{prefix}cell_first._ops.{off + j}.", O.FIRST_OPS, Wf[off + j], h, h_in)
                for j, h in enumerate(states))
        off += len(states)
        states.append(s)
    states = states[1:]
    states = [mixed_op(g, S, f"{prefix}cell_middle._ops.{i}.", O.MIDDLE_OPS, Wm[i], states[i], h_in)
              for i in range(nfirst)]                                 # Cell_Middle
    off = 0
    for _ in range(nlast):                                            # Cell_Last
        s = sum(mixed_op(g, S, f"{prefix}cell_last._ops.{off + j}.", O.LAST_OPS, Wl[off + j], h, h_in)
                for j, h in enumerate(states))
        off += len(states)
        states.append(s)
    return F.linear(torch.cat(states, dim=1), S[prefix + "concat_weights.weight"], S[prefix + "concat_weights.bias"])


def supernet_forward(g, S, alphas, node_id, src_in, edge_type, num_rel_rows, layers,
                     nzero=1, nfirst=2, nlast=2):
    """Reference models/model_search_lp.py:131-163 (+ show_weights :196-213)."""
    ent_all, rel_emb = _embed_tables(S)
    n = g.n
    dev = src_in.device
    src_in_f = torch.cat((src_in, torch.arange(n, device=dev)))
    src_id_f = node_id.view(-1)[src_in_f]
    et_f = torch.cat((edge_type, torch.full((n,), num_rel_rows - 1, dtype=torch.long, device=dev)))
    nfe = sum(nzero + i for i in range(nfirst))
    nle = sum(nfirst + i for i in range(nlast))
    ent = None
    for i in range(layers):
        Wz = F.softmax(alphas[0][i * nzero:(i + 1) * nzero], dim=1)
        Wf = F.softmax(alphas[1][i * nfe:(i + 1) * nfe], dim=1)
        Wm = F.softmax(alphas[2][i * nfirst:(i + 1) * nfirst], dim=1)
        Wl = F.softmax(alphas[3][i * nle:(i + 1) * nle], dim=1)
        x = ent_all[src_id_f] if i == 0 else torch.cat((ent[src_in], ent), dim=0)
        ent = super_cell(g, S, f"cells.{i}.", nfirst, nlast, x, rel_emb[et_f], Wz, Wf, Wm, Wl)
        ent = _bn(S, "batchnorm_h.", ent)
        if i > 0 or layers == 1:                      # model_search_lp.py:147-148,156
            ent = _relu(("net", i), ent)
        rel_emb = torch.matmul(rel_emb, S["w_rel"])
    return ent, rel_emb


def distmult_bce(ent, rel_emb, triplets, labels):
    """Reference models/model_search_lp.py:169-188."""
    s, r, o = ent[triplets[:, 0]], rel_emb[triplets[:, 1]], ent[triplets[:, 2]]
    return F.binary_cross_entropy_with_logits(torch.sum(s * r * o, dim=1), labels)
