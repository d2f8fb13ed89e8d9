"""f3 / f4 on the device, through the C ABI: negative sampling, node relabelling, the sampled
step graph, dense (label-smoothed) targets, filtered ranking and the [B, N] score functions.  Integer work is compared
BIT-EXACTLY with fixtures produced by running the reference itself (tests/golden/sampling_small.npz,
labels_ranking_small.npz; random draws replayed) and with the oracle restatement on larger inputs; the float score
functions against the reference's golden outputs / float64 within 1e-4."""
import numpy as np
import pytest
import torch

from conftest import load_golden
import mr_gnas_amd
from mr_gnas_amd import evaluation as EV, functional as K, operations_lp as O, sampler as SM
from oracle import dataprep as OD

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_sampled_step_graph_matches_reference_draw_for_draw():
    z = load_golden("sampling_small")
    draws = {"edges": z["draw_edges"], "values": z["draw_values"], "choices": z["draw_choices"], "split": z["draw_split"]}
    g, uniq_v, src_o, rel, node_norm, samples, labels = SM.generate_sampled_graph_and_labels(
        z["triples"].to(DEV), z["sample"], 0.5, z["R"], z["neg"], z["Nall"], draws=draws)
    assert torch.equal(uniq_v.cpu(), z["uniq_v"]) and torch.equal(src_o.cpu(), z["src_o"]) and torch.equal(rel.cpu(), z["rel"])
    assert torch.equal(node_norm.cpu(), z["node_norm"])                       # float32, bit for bit
    assert torch.equal(samples.cpu(), z["samples"]) and torch.equal(labels.cpu(), z["labels"])
    s, d, _ = g.edges(form="all")
    assert torch.equal(s.cpu(), z["g_src"]) and torch.equal(d.cpu(), z["g_dst"])
    assert g.edata["norm"].shape == (g.num_edges(), 1)
    assert torch.equal(g.edata["norm"].view(-1).cpu(), z["node_norm"][z["g_dst"]] * z["node_norm"][z["g_src"]])
    s2, l2 = SM.negative_sampling(z["ns_pos"].to(DEV), z["ns_num_entity"], 3, z["ns_values"], z["ns_choices"])
    assert torch.equal(s2.cpu(), z["ns_samples"]) and torch.equal(l2.cpu(), z["ns_labels"])


@pytest.mark.parametrize("name", ["small", "dense"])
def test_neighborhood_sampler_matches_reference_draw_for_draw(name):
    """sample_edge_neighborhood (reference utils/utils_rgcn.py:30-71, `--edge_sampler neighbor`) in one launch of one persistent
    workgroup: with the reference's own draws replayed (the uniform of every vertex pick, every adjacency slot it tried) the picked
    edges and the whole sampled step graph are bit-identical with the reference's; `dense` picks two thirds of all triples
    (long rejection runs, exhausted vertices, the all-ones restart of the weights)."""
    z = load_golden("sampling_neighbor_" + name)
    tri = z["triples"].to(DEV)
    adj = SM.AdjIndex(z["Nall"], tri)
    edges = SM.sample_edge_neighborhood(adj, z["sample"], {"u_vertex": z["draw_u_vertex"], "tries": z["draw_tries"]})
    assert torch.equal(edges.cpu(), z["edges"].long())
    draws = {"u_vertex": z["draw_u_vertex"], "tries": z["draw_tries"], "values": z["draw_values"], "choices": z["draw_choices"], "split": z["draw_split"]}
    g, uniq_v, src_o, rel, node_norm, samples, labels = SM.generate_sampled_graph_and_labels(
        tri, z["sample"], 0.5, z["R"], z["neg"], z["Nall"], sampler="neighbor", draws=draws, adj=adj)
    assert torch.equal(uniq_v.cpu(), z["uniq_v"]) and torch.equal(src_o.cpu(), z["src_o"]) and torch.equal(rel.cpu(), z["rel"])
    assert torch.equal(node_norm.cpu(), z["node_norm"])
    assert torch.equal(samples.cpu(), z["samples"]) and torch.equal(labels.cpu(), z["labels"])
    s, d, _ = g.edges(form="all")
    assert torch.equal(s.cpu(), z["g_src"]) and torch.equal(d.cpu(), z["g_dst"])
    # replayed tries that run out / point outside a list are reported, not read
    with pytest.raises(RuntimeError):
        SM.sample_edge_neighborhood(adj, z["sample"], {"u_vertex": z["draw_u_vertex"], "tries": z["draw_tries"][:5]})


def test_neighborhood_sampler_with_device_draws_expands_a_neighbourhood():
    """Without replayed draws (torch's device generator, one draw per edge pick among the vertex's unpicked entries): picks are
    distinct triples, and every pick after a restart-free prefix touches a vertex that an earlier pick has seen -- the sample is
    connected, which is the point of the scheme; also through generate_sampled_graph_and_labels at the driver's sizes."""
    from mr_gnas_amd import synth
    n, r, t = synth.SHAPES["fb15k237"]
    tri = torch.from_numpy(synth.synth_kg(n, r, t, 0)).to(DEV)
    adj = SM.AdjIndex(n, tri)
    gen = torch.Generator(device=DEV).manual_seed(11)
    for sample in (300, 3000):
        e = SM.sample_edge_neighborhood(adj, sample, generator=gen)
        assert e.numel() == sample and torch.unique(e).numel() == sample and int(e.min()) >= 0 and int(e.max()) < t
        picked = tri[e].cpu().numpy()
        seen = {int(picked[0, 0]), int(picked[0, 2])}
        for s_, _, o_ in picked[1:]:
            assert int(s_) in seen or int(o_) in seen           # connected growth (FB15k-237's giant component: no restart)
            seen.update((int(s_), int(o_)))
    g, uniq_v, src_o, rel, node_norm, samples, labels = SM.generate_sampled_graph_and_labels(tri, 3000, 0.5, r, 10, n, sampler="neighbor",
                                                                                             generator=gen, adj=adj)
    assert samples.shape == (11 * 3000, 3) and g.num_edges() == 3000 and g.number_of_nodes() == int(uniq_v.numel())
    with pytest.raises(ValueError):
        SM.generate_sampled_graph_and_labels(tri, 300, 0.5, r, 10, n, sampler="random-walk")


def test_sampler_with_device_draws_has_the_reference_properties():
    """With torch's device generator (no numpy stream to replay): the scheme's invariants at the search driver's
    default and large sizes."""
    from mr_gnas_amd import synth
    n, r, t = synth.SHAPES["fb15k237"]
    tri = torch.from_numpy(synth.synth_kg(n, r, t, 0)).to(DEV)
    gen = torch.Generator(device=DEV).manual_seed(5)
    for sample in (300, 30000):
        g, uniq_v, src_o, rel, node_norm, samples, labels = SM.generate_sampled_graph_and_labels(tri, sample, 0.5, r, 10, n, generator=gen)
        nn_ = int(uniq_v.numel())
        assert bool((uniq_v[1:] > uniq_v[:-1]).all()) and int(uniq_v.max()) < n
        assert samples.shape == (11 * sample, 3) and float(labels.sum()) == sample and bool((labels[:sample] == 1).all())
        pos, neg = samples[:sample], samples[sample:]
        assert int(pos[:, [0, 2]].max()) < nn_ and int(neg[:, [0, 2]].max()) < nn_
        tiled = pos.repeat(10, 1)
        changed_s, changed_o = neg[:, 0] != tiled[:, 0], neg[:, 2] != tiled[:, 2]
        assert bool((neg[:, 1] == tiled[:, 1]).all()) and not bool((changed_s & changed_o).any())   # exactly one end is corrupted
        assert 0.4 < float(changed_s.float().mean()) < 0.6
        # positives are the relabelled picked triples: mapping back through uniq_v gives rows of the KG
        back = torch.stack((uniq_v[pos[:, 0]], pos[:, 1], uniq_v[pos[:, 2]]), 1)
        key = lambda x: (x[:, 0] * (2 * r) + x[:, 1]) * n + x[:, 2]
        assert bool(torch.isin(key(back), key(tri)).all())
        assert g.num_edges() == 2 * int(sample * 0.5) and g.number_of_nodes() == nn_
        et = g.edata["e_type"]
        s, d, _ = g.edges(form="all")
        k2 = (et * nn_ + d) * nn_ + s
        assert bool((k2[1:] >= k2[:-1]).all())                                  # sorted by (relation, dst, src)
        # against the oracle with the same picks (recovered from the outputs): relabel + graph agree
        og = OD.build_search_graph(nn_, r, torch.stack((s[: g.num_edges() // 2], et[: g.num_edges() // 2], d[: g.num_edges() // 2]), 1).cpu().numpy())
        assert torch.equal(og.src, s.cpu()) and torch.equal(og.norm.view(-1), g.edata["norm"].view(-1).cpu())


def test_relabel_nodes_is_np_unique():
    rng = np.random.default_rng(0)
    for n, num_nodes in ((1, 10), (500, 100), (5000, 1 << 20), (7, 7)):
        a, b = rng.integers(0, num_nodes, n), rng.integers(0, num_nodes, n)
        uv, inv = np.unique((a, b), return_inverse=True)
        u, na, nb = SM.relabel_nodes(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV), num_nodes)
        assert np.array_equal(u.cpu().numpy(), uv)
        assert np.array_equal(torch.stack((na, nb)).cpu().numpy(), inv.reshape(2, -1))


def test_labels_and_filtered_ranking_match_reference():
    z = load_golden("labels_ranking_small")
    N, R = z["N"], z["R"]
    train_idx = SM.LabelIndex(z["train"], R, N, DEV)
    all_idx = SM.LabelIndex(torch.cat((z["train"], z["valid"], z["test"])), R, N, DEV)
    t = z["train_triples"].to(DEV)
    assert torch.equal(train_idx.labels(t[:, 0], t[:, 1], 0.1).cpu(), z["train_labels"])       # label smoothing, float32 bit for bit
    t = z["test_triples"].to(DEV)
    lab = all_idx.labels(t[:, 0], t[:, 1])
    assert torch.equal(lab.cpu(), z["test_labels"])
    ranks = EV.filtered_ranks(z["pred"].to(DEV), lab, t[:, 2])
    assert torch.equal(ranks.cpu(), z["ranks"])

    class Prepared:                                                            # predict() with a model returning the fixture's scores
        def __init__(self, rows):
            self.rows = rows

        def eval(self):
            pass

        def __call__(self, g, subj, rel):
            return self.rows.pop(0)

    bs = 16
    loader = [(z["test_triples"][i:i + bs], z["test_labels"][i:i + bs]) for i in range(0, len(z["ranks"]), bs)]
    res, loss = EV.predict(loader, None, Prepared([z["pred"][i:i + bs].to(DEV) for i in range(0, len(z["ranks"]), bs)]), DEV)
    assert res["count"] == int(z["res/count"]) and res["mr"] == float(z["res/mr"])
    assert abs(res["mrr"] - float(z["res/mrr"])) <= 1e-4 * float(z["res/mrr"])
    for k in (1, 3, 10):
        assert res[f"hits@{k}"] == int(z[f"res/hits@{k}"])
    assert abs(loss - float(z["res_loss"])) <= 1e-4 * float(z["res_loss"])
    comb = EV.combine_results(res, res)
    assert comb["mrr"] == round(res["mrr"] / res["count"], 5)


def test_filtered_ranking_full_size_and_ties():
    """FB15k-237-sized rows (N = 14 541, B = 256) against the oracle's argsort formulation; exact ties (saturated
    sigmoids) follow the stable-sort convention."""
    gen = torch.Generator().manual_seed(3)
    B, N = 256, 14541
    pred = torch.sigmoid(torch.randn(B, N, generator=gen) * 6)                  # many scores saturate to exactly 1.0f / tie
    labels = (torch.rand(B, N, generator=gen) < 0.002).float()
    obj = torch.randint(0, N, (B,), generator=gen)
    labels[torch.arange(B), obj] = 1
    want = OD.filtered_ranks(pred, labels, obj)
    got = EV.filtered_ranks(pred.to(DEV), labels.to(DEV), obj.to(DEV))
    assert torch.equal(got.cpu(), want)
    assert int((pred == 1.0).sum()) > 100                                      # the tie rule was exercised


@pytest.mark.parametrize("case", ["ops_tiny_train", "ops_small_search", "ops_odd_train"])
def test_score_functions_against_reference_golden(case):
    z = load_golden(case)
    for nm in ("sf_DisMult", "sf_TransE"):
        op = O.MIXED_OPS_sf[nm]({"gamma": 9.0})
        ent = z["xn"].to(DEV).requires_grad_(True)
        s = z[nm + "/sub"].to(DEV).requires_grad_(True)
        r = z[nm + "/rel"].to(DEV).requires_grad_(True)
        out = op(ent, s, r)
        out.backward(z[nm + "/gout"].to(DEV))
        for got, key in ((out, "/out"), (ent.grad, "/gent"), (s.grad, "/gsub"), (r.grad, "/grel")):
            ref = z[nm + key]
            err = float((got.detach().cpu() - ref).abs().max())
            assert err <= 1e-4 * max(1.0, float(ref.abs().max())), f"{case} {nm}{key}: {err:.3e}"


def test_score_functions_full_size_against_float64():
    """[256, 14 541] scores at D = 200 (the fixed-genotype driver's batch, reference train/mr_lp_train.py:225-240)."""
    gen = torch.Generator(device=DEV).manual_seed(9)
    B, N, D = 256, 14541, 200
    ent = torch.randn(N, D, device=DEV, generator=gen) * 0.3
    sub = torch.randn(B, D, device=DEV, generator=gen) * 0.3
    rel = torch.randn(B, D, device=DEV, generator=gen) * 0.3
    gup = torch.randn(B, N, device=DEV, generator=gen)
    for nm, gamma in (("sf_DisMult", None), ("sf_TransE", 40.0)):
        e, s, r = (t.clone().requires_grad_(True) for t in (ent, sub, rel))
        out = O.MIXED_OPS_sf[nm]({"gamma": gamma})(e, s, r)
        out.backward(gup)
        e64, s64, r64 = (t.double().clone().requires_grad_(True) for t in (ent, sub, rel))
        if nm == "sf_DisMult":
            ref = torch.sigmoid((s64 * r64) @ e64.t())
        else:
            ref = torch.sigmoid(gamma - torch.cdist(s64 + r64, e64, p=1))
        ref.backward(gup.double())
        assert float((out.detach().double() - ref.detach()).abs().max()) <= 1e-4, nm
        for got, want, what in ((e.grad, e64.grad, "ent"), (s.grad, s64.grad, "sub"), (r.grad, r64.grad, "rel")):
            err = float((got.double() - want).abs().max())
            assert err <= 2e-4 * max(1.0, float(want.abs().max())), f"{nm} grad {what}: {err:.3e}"


@pytest.mark.parametrize("sample,negative", [(300, 10), (37, 1), (5000, 3)])
def test_static_step_draws_what_the_host_driven_sampler_draws(sample, negative):
    """sampler.static_step (no host read, capacity-padded shapes, node count on the device: the search step as one replayable HIP
    graph) against generate_sampled_graph_and_labels (reference utils/utils_rgcn.py:79-118) on the SAME random stream: the same edge
    pick, the same relabelling (node ids of the draw, padding entries 0 behind them), the same corrupted triples and labels, the same
    graph split -- edge for edge."""
    import torch
    from mr_gnas_amd import sampler as SM
    DEV = "cuda"
    gen0 = torch.Generator().manual_seed(sample)
    N_all, R, T = 4000, 9, 60000
    tri = torch.stack((torch.randint(0, N_all, (T,), generator=gen0), torch.randint(0, R, (T,), generator=gen0),
                       torch.randint(0, N_all, (T,), generator=gen0)), 1).to(DEV)
    ga, gb = torch.Generator(device=DEV).manual_seed(77), torch.Generator(device=DEV).manual_seed(77)
    st = SM.static_step(tri, sample, 0.5, R, negative, N_all, generator=ga)
    g, uniq_v, src_o, rel, node_norm, samples, labels = SM.generate_sampled_graph_and_labels(tri, sample, 0.5, R, negative, N_all, generator=gb)
    n = int(st["n_nodes"].item())
    assert n == uniq_v.numel() and st["cap"] == min(2 * sample, N_all) and int(st["n_rows"].item()) == n + g.num_edges()
    assert torch.equal(st["node_id"].view(-1)[:n], uniq_v.view(-1)) and int(st["node_id"].view(-1)[n:].abs().max() if n < st["cap"] else 0) == 0
    # positives and labels are the same; the corrupted entities are drawn as floor(rand * n) against a DEVICE n (the host-driven
    # sampler draws randint(n) against a host n: another stream, the same distribution) -- checked structurally
    B = sample
    assert torch.equal(st["samples"][:B], samples[:B]) and torch.equal(st["labels"], labels)
    neg = st["samples"][B:].view(negative, B, 3) if negative else st["samples"][B:].view(0, B, 3)
    pos = st["samples"][:B].unsqueeze(0).expand_as(neg)
    assert int(st["samples"][:, [0, 2]].min()) >= 0 and int(st["samples"][:, [0, 2]].max()) < n
    assert torch.equal(neg[..., 1], pos[..., 1])                                     # the relation is never corrupted
    same_head, same_tail = neg[..., 0] == pos[..., 0], neg[..., 2] == pos[..., 2]
    assert bool((same_head | same_tail).all())                                       # exactly one end is replaced (it may draw the same id)
    if B * negative >= 3000:                                                         # uniform over [0, n): mean and head / tail balance
        drawn = torch.where(same_tail, neg[..., 0], neg[..., 2]).double()
        assert abs(float(drawn.mean()) - (n - 1) / 2) <= 0.05 * n and 0.4 <= float((~same_head).double().mean()) <= 0.6
    assert st["g"].num_edges() == g.num_edges() and st["g"].number_of_nodes() == st["cap"]
    for a, b in zip(st["g"].edges(form="all")[:2], g.edges(form="all")[:2]):
        assert torch.equal(a, b)
    assert torch.equal(st["src"], src_o) and torch.equal(st["rel"], rel)
    # the edge norm of the padded graph on the draw's edges is the unpadded graph's (isolated padding nodes change no in-degree)
    assert torch.equal(st["g"].edata["norm"], g.edata["norm"])
