"""The CPU oracle against the golden vectors produced by running the reference
(tests/golden/make_golden.py).  CPU only."""
import collections

import numpy as np
import pytest
import torch

from conftest import assert_param_grad, load_golden, net_grad_names, net_params, ops_inputs, sub
from oracle import compgcn as OC
from oracle import nets as ON
from oracle import ops as OO
from oracle.graph import OGraph, build_search_graph, build_train_graph

OPS_CASES = ["ops_tiny_train", "ops_small_search", "ops_mid_train", "ops_d100_search", "ops_odd_train", "ops_r300_d64_search"]
TOL = dict(rtol=2e-5, atol=2e-6)


def graph_of(z):
    return OGraph(z["N"], z["src"], z["dst"], z["etype"], z["norm"])


def run(fn, g, P, a, b, gout):
    a = a.clone().requires_grad_(True)
    b = b.clone().requires_grad_(True)
    P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    out = fn(g, P, a, b)
    out.backward(gout)
    z = torch.zeros_like
    return out, (a.grad if a.grad is not None else z(a)), (b.grad if b.grad is not None else z(b)), \
        {k: (v.grad if v.grad is not None else z(v)) for k, v in P.items()}


@pytest.mark.parametrize("case", OPS_CASES)
def test_ops_match_reference(case):
    z = load_golden(case)
    g = graph_of(z)
    tags = sorted({k.split("/")[0] for k in z if k.endswith("/out") and not k.startswith("sf_")})
    assert "f_sparse_comp" in tags and "a_max" in tags
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
        out, ga, gb, gp = run(OO.OPS[name], g, sub(z, tag + "/param/"), a, b, gout)
        torch.testing.assert_close(out, z[tag + "/out"], **TOL, msg=lambda m: f"{case}:{tag} out {m}")
        torch.testing.assert_close(ga, z[tag + "/ga"], **TOL, msg=lambda m: f"{case}:{tag} ga {m}")
        torch.testing.assert_close(gb, z[tag + "/gb"], **TOL, msg=lambda m: f"{case}:{tag} gb {m}")
        for k, v in gp.items():
            torch.testing.assert_close(v, z[f"{tag}/gparam/{k}"], rtol=1e-4, atol=1e-5,
                                       msg=lambda m: f"{case}:{tag} gparam {k} {m}")


@pytest.mark.parametrize("case", ["ops_tiny_train", "ops_small_search", "ops_odd_train"])
def test_score_functions(case):
    z = load_golden(case)
    for nm in ("sf_DisMult", "sf_TransE"):
        ent = z["xn"].clone().requires_grad_(True)
        s = z[nm + "/sub"].clone().requires_grad_(True)
        r = z[nm + "/rel"].clone().requires_grad_(True)
        out = OO.SF[nm](ent, s, r, 9.0)
        out.backward(z[nm + "/gout"])
        torch.testing.assert_close(out, z[nm + "/out"], **TOL)
        torch.testing.assert_close(ent.grad, z[nm + "/gent"], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(s.grad, z[nm + "/gsub"], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(r.grad, z[nm + "/grel"], rtol=1e-4, atol=1e-5)


def test_graph_builders_bit_exact():
    z = load_golden("graph_small")
    for order, fn in (("train", build_train_graph), ("search", build_search_graph)):
        g = fn(z["N"], z["R"], z["triples"].numpy())
        assert torch.equal(g.src, z[order + "/src"])
        assert torch.equal(g.dst, z[order + "/dst"])
        assert torch.equal(g.etype, z[order + "/etype"])
        assert torch.equal(g.norm, z[order + "/norm"]), order      # float32 bit-exact
    for case, fn in (("ops_tiny_train", build_train_graph), ("ops_small_search", build_search_graph)):
        z = load_golden(case)
        g = fn(z["N"], z["R"], z["triples"].numpy())
        assert torch.equal(g.src, z["src"]) and torch.equal(g.dst, z["dst"]) and torch.equal(g.etype, z["etype"])
        assert torch.equal(g.norm, z["norm"])


def test_compgcn_layer_and_stack():
    z = load_golden("compgcn_small")
    g = OGraph(z["N"], z["src"], z["dst"], z["etype"], z["norm"])
    in_mask = z["in_edges_mask"].bool()
    for fn_ in ("sub", "mul", "ccorr"):
        for bnorm in (True, False):
            tag = f"conv_{fn_}_{'bn' if bnorm else 'nobn'}"
            for corr in ((OC.ccorr, OC.ccorr_direct) if fn_ == "ccorr" else (OC.ccorr,)):
                P = {k: v.clone().requires_grad_(True) for k, v in sub(z, tag + "/param/").items()}
                a = z["n_in"].clone().requires_grad_(True)
                b = z["r_in"].clone().requires_grad_(True)
                no, ro = OC.comp_graph_conv(g, P, a, b, in_mask, fn_, bnorm, corr=corr)
                ((no * z["gn"]).sum() + (ro * z["gr"]).sum()).backward()
                tol = dict(rtol=1e-4, atol=2e-5)
                torch.testing.assert_close(no, z[tag + "/n_out"], **tol)
                torch.testing.assert_close(ro, z[tag + "/r_out"], **tol)
                torch.testing.assert_close(a.grad, z[tag + "/gn_in"], **tol)
                torch.testing.assert_close(b.grad, z[tag + "/gr_in"], **tol)
                for k, v in P.items():
                    torch.testing.assert_close(v.grad, z[f"{tag}/gparam/{k}"], rtol=2e-4, atol=5e-5, msg=lambda m: f"{tag} {k} {m}")
    for tag, fn_ in (("net_sub_b3", "sub"), ("net_mul_b0", "mul")):
        P = {k: v.clone().requires_grad_(True) for k, v in sub(z, tag + "/param/").items()}
        no, ro = OC.comp_gcn(g, P, in_mask, 2, fn_)
        ((no * z[tag + "/go_n"]).sum() + (ro * z[tag + "/go_r"]).sum()).backward()
        torch.testing.assert_close(no, z[tag + "/n_out"], rtol=1e-4, atol=2e-5)
        torch.testing.assert_close(ro, z[tag + "/r_out"], rtol=1e-4, atol=2e-5)
        for k, v in P.items():
            torch.testing.assert_close(v.grad, z[f"{tag}/gparam/{k}"], rtol=5e-4, atol=5e-5, msg=lambda m: f"{tag} {k} {m}")


Genotype = collections.namedtuple("Genotype", "alpha_cell concat_node score_func")   # reference configs/genotypes.py:3
README_GENOTYPE = [Genotype(alpha_cell=[('pre_sub', 1, 0), ('f_sparse_comp', 2, 1), ('f_sparse_comp', 3, 2),
                                        ('a_max', 4, 2), ('a_max', 5, 3), ('f_sparse_last', 6, 5),
                                        ('f_sparse_last', 7, 5)], concat_node=[4, 5, 6, 7], score_func='sf_DisMult')]


@pytest.mark.parametrize("case", ["fixednet_tiny", "fixednet_d64"])
def test_fixed_genotype_network(case):
    z = load_golden(case)
    g = OGraph(z["N"], z["src"], z["dst"], z["etype"], z["norm"])
    S = {k: v.clone().requires_grad_(True) for k, v in sub(z, "param/").items()}
    pred = ON.fixed_net_forward(g, S, README_GENOTYPE, z["subj"], z["rel"], 2 * z["R"] + 1, gamma=9.0)
    loss = torch.nn.functional.binary_cross_entropy(pred, z["label"])
    loss.backward()
    torch.testing.assert_close(pred, z["pred"], rtol=1e-4, atol=5e-5)
    torch.testing.assert_close(loss.detach(), z["loss"], rtol=1e-5, atol=1e-6)
    for k, v in S.items():
        gref = z["gparam/" + k]
        got = v.grad if v.grad is not None else torch.zeros_like(v)
        scale = max(float(gref.abs().max()), 1e-6)
        assert float((got - gref).abs().max()) <= 2e-4 * scale + 2e-6, k


@pytest.mark.parametrize("case", ["supernet_tiny", "supernet_d24", "supernet_d200_sampled"])
def test_supernet_step(case):
    torch.set_num_threads(1)      # as the generator ran the reference: at D = 200 the thread count changes reduction orders
    z = load_golden(case)
    n = z["node_id"].numel()
    g = OGraph(n, z["src"], z["dst"], z["edge_type"], z["norm"])
    S = {k: v.clone().requires_grad_(True) for k, v in net_params(z).items()}
    alphas = [z[f"alpha/{i}"].clone().requires_grad_(True) for i in range(5)]
    ent, rel = ON.supernet_forward(g, S, alphas, z["node_id"], z["src_in"], z["edge_type"], 2 * z["R"] + 1, z["layers"])
    loss = ON.distmult_bce(ent, rel, z["data"], z["labels"])
    loss.backward()
    torch.testing.assert_close(ent, z["ent"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rel, z["rel_out"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(loss.detach(), z["loss"], rtol=1e-5, atol=1e-6)
    for i in range(4):
        gref = z[f"galpha/{i}"]
        scale = max(float(gref.abs().max()), 1e-8)
        assert float((alphas[i].grad - gref).abs().max()) <= 1e-3 * scale + 1e-8, f"alpha {i}"
    assert sorted(S) == net_grad_names(z)
    for k, v in S.items():
        assert_param_grad(z, k, v.grad if v.grad is not None else torch.zeros_like(v), 1e-3, 2e-6, case)


def test_data_preparation_oracle_matches_reference():
    """oracle/dataprep.py against what the reference's own functions produced (tests/golden/make_golden.py:
    case_sampling, case_labels_and_ranking)."""
    from oracle import dataprep as OD
    z = load_golden("sampling_small")
    draws = {"edges": z["draw_edges"].numpy(), "values": z["draw_values"].numpy(), "choices": z["draw_choices"].numpy(),
             "split": z["draw_split"].numpy()}
    g, uniq_v, src_o, rel, node_norm, samples, labels = OD.sampled_graph_and_labels(z["triples"].numpy(), z["sample"], 0.5, z["R"], z["neg"], draws)
    assert np.array_equal(uniq_v, z["uniq_v"].numpy()) and np.array_equal(src_o, z["src_o"].numpy()) and np.array_equal(rel, z["rel"].numpy())
    assert np.array_equal(node_norm, z["node_norm"].numpy())
    assert np.array_equal(samples, z["samples"].numpy()) and np.array_equal(labels, z["labels"].numpy())
    assert torch.equal(g.src, z["g_src"]) and torch.equal(g.dst, z["g_dst"])
    s2, l2 = OD.negative_sampling(z["ns_pos"].numpy(), z["ns_num_entity"], 3, z["ns_values"].numpy(), z["ns_choices"].numpy())
    assert np.array_equal(s2, z["ns_samples"].numpy()) and np.array_equal(l2, z["ns_labels"].numpy())

    for name in ("small", "dense"):                 # the neighbourhood-expansion sampler (utils_rgcn.py:30-71), draws replayed
        z = load_golden("sampling_neighbor_" + name)
        adj, deg = OD.adjacency(z["Nall"], z["triples"].numpy())
        edges = OD.sample_edge_neighborhood(adj, deg, len(z["triples"]), z["sample"], z["draw_u_vertex"].numpy(), z["draw_tries"].numpy())
        assert np.array_equal(edges, z["edges"].numpy())
        draws = {"edges": edges, "values": z["draw_values"].numpy(), "choices": z["draw_choices"].numpy(), "split": z["draw_split"].numpy()}
        g, uniq_v, src_o, rel, node_norm, samples, labels = OD.sampled_graph_and_labels(z["triples"].numpy(), z["sample"], 0.5, z["R"], z["neg"], draws)
        assert np.array_equal(uniq_v, z["uniq_v"].numpy()) and np.array_equal(samples, z["samples"].numpy()) and torch.equal(g.src, z["g_src"])

    z = load_golden("labels_ranking_small")
    tr = OD.sr2o(z["train"].numpy(), z["R"])
    al = OD.sr2o(torch.cat((z["train"], z["valid"], z["test"])).numpy(), z["R"])
    t = z["train_triples"]
    assert torch.equal(OD.dense_labels(tr, t[:, 0], t[:, 1], z["N"], 0.1), z["train_labels"])
    t = z["test_triples"]
    assert torch.equal(OD.dense_labels(al, t[:, 0], t[:, 1], z["N"]), z["test_labels"])
    ranks = OD.filtered_ranks(z["pred"], z["test_labels"], t[:, 2])
    assert torch.equal(ranks, z["ranks"])
    assert int(z["res/count"]) == len(ranks) and float(z["res/mr"]) == float(ranks.sum())
    assert abs(float(z["res/mrr"]) - float((1.0 / ranks.float()).sum())) < 1e-3
    for k in (1, 3, 10):
        assert int(z[f"res/hits@{k}"]) == int((ranks <= k).sum())
