"""Search-step sampling, negative sampling and label construction on the device -- the data preparation around the
hot path that the reference does in numpy on the host for every step:

* ``generate_sampled_graph_and_labels`` / ``sample_edge_uniform`` / ``negative_sampling``
  (reference utils/utils_rgcn.py:73-118, 191-204): same names, same return values, tensors instead of numpy arrays;
* ``LabelIndex`` (reference utils/process_data.py:4-31 + utils/data_set.py:6-59): the ``sr2o`` dictionaries as a
  sorted key / CSR pair in HBM and the dense (label-smoothed) ``[B, num_ent]`` targets of a batch.

Everything integer runs in the HIP kernels of ``csrc/sampling.hip`` / ``csrc/plans.hip`` behind the C ABI.  Random
draws come from torch's device generator (the reference uses numpy's global generator; bit-equal streams are not
possible across generators), but every function takes the draws as optional arguments and is bit-exact with the
reference for the same draws -- that is what the tests replay.  The neighbourhood-expansion sampler
(``sample_edge_neighborhood``, utils_rgcn.py:30-71, ``--edge_sampler neighbor``) is ``sample_size`` dependent picks: one
launch of one persistent workgroup (``mrg_sample_edge_neighborhood``) over an adjacency CSR in the reference's append order.
"""
import numpy as np
import torch

from . import graph as G
from ._lib import call, load, ptr, require_hip, stream_of


def sample_edge_uniform(n_triplets, sample_size, device, generator=None):
    """`np.random.choice(n_triplets, sample_size, replace=False)` (reference utils/utils_rgcn.py:73-76)."""
    return torch.randperm(int(n_triplets), device=device, generator=generator)[: int(sample_size)]


class AdjIndex:
    """``get_adj_and_degrees`` (reference utils/utils_rgcn.py:18-28) resident in HBM: the adjacency lists in the reference's
    append order (for triple i: ``[i, o]`` joins s's list, then ``[i, s]`` joins o's) as a CSR, plus the degrees."""

    def __init__(self, num_nodes, triplets):
        t = triplets.long()
        T, dev = int(t.shape[0]), t.device
        self.N, self.T = int(num_nodes), T
        vert = torch.stack((t[:, 0], t[:, 2]), dim=1).reshape(-1)            # entry 2i: s's list, entry 2i + 1: o's list
        other = torch.stack((t[:, 2], t[:, 0]), dim=1).reshape(-1)
        eid = torch.arange(T, device=dev).repeat_interleave(2)
        order = torch.sort(vert, stable=True).indices                        # stable: append order inside every list
        self.adj_edge = eid[order].to(torch.int32).contiguous()
        self.adj_other = other[order].to(torch.int32).contiguous()
        deg = torch.bincount(vert, minlength=self.N)
        self.degrees = deg.to(torch.int32).contiguous()
        rowptr = torch.zeros(self.N + 1, dtype=torch.int64, device=dev)
        rowptr[1:] = torch.cumsum(deg, 0)
        self.rowptr = rowptr.to(torch.int32).contiguous()
        G.settle(dev)


def sample_edge_neighborhood(adj, sample_size, draws=None, generator=None):
    """Reference utils/utils_rgcn.py:30-71: `sample_size` edges by neighbourhood expansion (the sampled edges form a connected
    graph).  `adj`: AdjIndex.  `draws`: dict(u_vertex=float64 [sample_size], tries=int64 [n]) replays the reference's own draws
    (the uniform behind every `np.random.choice(..., p=...)` and the sequence of adjacency slots it tried, rejected ones
    included): bit-exact.  Without draws the uniforms come from torch's device generator and the edge of a vertex is ONE draw
    among its unpicked entries (the reference's distribution without its rejection loop).  Returns int64 edge ids."""
    dev = adj.rowptr.device
    S = int(sample_size)
    draws = draws or {}
    u_vertex = draws.get("u_vertex")
    u_vertex = (torch.rand(S, dtype=torch.float64, device=dev, generator=generator) if u_vertex is None
                else torch.as_tensor(u_vertex).to(dev).double().contiguous())
    tries = draws.get("tries")
    u_edge = None
    if tries is None:
        u_edge = torch.rand(S, dtype=torch.float64, device=dev, generator=generator)
    else:
        tries = torch.as_tensor(tries).to(dev).long().contiguous()
    if u_vertex.numel() != S:
        raise ValueError(f"sample_edge_neighborhood needs {S} vertex draws, got {u_vertex.numel()}")
    require_hip(adj.rowptr, u_vertex, tries, u_edge)
    edges = torch.empty(S, dtype=torch.int32, device=dev)
    status = torch.zeros(3, dtype=torch.int64, device=dev)
    nb = load().mrg_sample_neighborhood_workspace_bytes(adj.N, adj.T)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
    call("mrg_sample_edge_neighborhood", (ptr(adj.rowptr), ptr(adj.adj_edge), ptr(adj.adj_other), ptr(adj.degrees), adj.N, adj.T, S, ptr(u_vertex),
                                          ptr(tries), int(tries.numel()) if tries is not None else 0, ptr(u_edge), ptr(edges), ptr(status),
                                          ptr(ws), nb, stream_of(edges)))
    made, used, code = (int(v) for v in status.tolist())
    if code != 0 or made != S:
        raise RuntimeError({1: "sample_edge_neighborhood: the graph has fewer incident edges than sample_size",
                            2: "sample_edge_neighborhood: a replayed adjacency slot is out of range",
                            3: "sample_edge_neighborhood: the replayed tries ran out"}.get(code, f"sample_edge_neighborhood failed (code {code})")
                           + f" after {made} of {S} picks")
    return edges.long()


def negative_sampling(pos_samples, num_entity, negative_rate, values=None, choices=None, generator=None):
    """Reference utils/utils_rgcn.py:191-204.  pos_samples [B, 3] int64 on the device; returns (samples
    [(rate+1) B, 3] int64, labels float32).  `values` (int64 [B * rate] in [0, num_entity)) and `choices` (float64
    [B * rate] in [0, 1)) are the reference's two random draws; drawn on the device when absent.  `num_entity` may be a DEVICE
    tensor (the node count of a static step graph, static_step below): the bound of the drawn entity ids then never visits the host."""
    pos = pos_samples.long().contiguous()
    require_hip(pos)
    B, n = int(pos.shape[0]), int(pos.shape[0]) * int(negative_rate)
    dev = pos.device
    if values is None and torch.is_tensor(num_entity):
        ne = num_entity.to(torch.float64).reshape(())
        values = torch.minimum((torch.rand(n, device=dev, dtype=torch.float64, generator=generator) * ne).floor(), ne - 1).clamp_(min=0).long()
    elif values is None:
        values = torch.randint(0, int(num_entity), (n,), device=dev, generator=generator)
    if choices is None:
        choices = torch.rand(n, device=dev, dtype=torch.float64, generator=generator)
    values, choices = values.to(dev).long().contiguous(), choices.to(dev).double().contiguous()
    if values.numel() != n or choices.numel() != n:
        raise ValueError(f"negative_sampling needs {n} draws, got {values.numel()} / {choices.numel()}")
    samples = torch.empty(B * (negative_rate + 1), 3, dtype=torch.int64, device=dev)
    labels = torch.empty(B * (negative_rate + 1), dtype=torch.float32, device=dev)
    call("mrg_negative_sampling", (ptr(pos), B, int(negative_rate), ptr(values), ptr(choices), ptr(samples), ptr(labels), stream_of(pos)))
    return samples, labels


def relabel_nodes(src, dst, num_nodes, static=False):
    """`uniq_v, edges = np.unique((src, dst), return_inverse=True)` (reference utils/utils_rgcn.py:97-101):
    returns (uniq_v ascending, new_src, new_dst).  static: no host read -- returns (uniq_v padded with node 0 to its host-known
    capacity min(2 n, num_nodes), new_src, new_dst, count [1] int32 on the device)."""
    src, dst = src.long().contiguous(), dst.long().contiguous()
    require_hip(src, dst)
    dev, n = src.device, int(src.numel())
    cap = max(min(2 * n, int(num_nodes)), 1)
    uniq = torch.zeros(cap, dtype=torch.int64, device=dev) if static else torch.empty(cap, dtype=torch.int64, device=dev)
    ns, nd = torch.empty_like(src), torch.empty_like(dst)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    nb = load().mrg_relabel_workspace_bytes(int(num_nodes))
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
    call("mrg_relabel_nodes", (ptr(src), ptr(dst), n, int(num_nodes), ptr(uniq), ptr(ns), ptr(nd), ptr(count), ptr(ws), nb, stream_of(src)))
    if static:
        return uniq, ns, nd, count
    return uniq[: int(count.item())], ns, nd


def static_step(triplets, sample_size, split_size, num_rels, negative_rate, num_nodes, generator=None):
    """generate_sampled_graph_and_labels (uniform sampler) WITHOUT a host read and with host-known shapes, so that a whole search step
    -- this function included -- can be captured into one HIP graph and replayed with a new draw every time (reference
    search/mr_lp_search.py:187-245: a new step graph per step; utils/utils_rgcn.py:79-118).  The number of distinct nodes of a draw
    stays in device memory: the step graph has `cap = min(2 * sample_size, num_nodes)` nodes, of which the first *n_nodes are the
    draw's (relabelled 0 .. n - 1 as the reference does) and the rest are isolated padding nodes (uniq_v entry 0; no edge touches
    them; the MixedOp kernels keep their rows zero and out of every statistic: mrg_set_dynamic_rows, supernet.SearchNetwork.static_rows).
    Draws come from torch's default device generator unless `generator` is registered with the capturing graph.
    Returns dict(g, node_id [cap, 1], src, rel, samples, labels, n_nodes [1] int32, n_rows [1] int32 = E + n_nodes, cap)."""
    dev = triplets.device
    T = int(triplets.shape[0])
    pick = sample_edge_uniform(T, sample_size, dev, generator)
    edges = triplets[pick].long()
    uniq_v, src, dst, count = relabel_nodes(edges[:, 0], edges[:, 2], num_nodes, static=True)
    rel = edges[:, 1].contiguous()
    relabeled = torch.stack((src, rel, dst), dim=1)
    samples, labels = negative_sampling(relabeled, count, negative_rate, generator=generator)
    n_split = int(sample_size * split_size)
    split = torch.randperm(int(sample_size), device=dev, generator=generator)[:n_split]
    cap = int(uniq_v.numel())
    graph_triples = relabeled[split]
    g = G.build_search_graph(cap, num_rels, graph_triples, device=dev)
    src_o, _, _ = g.edges(form="all")
    n_rows = count + int(g.num_edges())
    return dict(g=g, node_id=uniq_v.view(-1, 1), src=src_o, rel=g.edata["e_type"], samples=samples, labels=labels,
                n_nodes=count, n_rows=n_rows.to(torch.int32), cap=cap, graph_triples=graph_triples)


def generate_sampled_graph_and_labels(triplets, sample_size, split_size, num_rels, negative_rate, num_nodes, sampler="uniform",
                                      draws=None, generator=None, adj=None):
    """One search-step sample (reference utils/utils_rgcn.py:79-118).  `triplets` [T, 3] int64 on the device.
    Returns ``(g, uniq_v, src, rel, node_norm, samples, labels)`` like the reference, with ``g`` a RelGraph that
    already carries ``edata['e_type']`` and the edge norm ``edata['norm']`` [E, 1] (what the search driver computes
    next with node_norm_to_edge_norm, search/mr_lp_search.py:30-36,214), everything resident in HBM.
    `draws`: optional dict(edges, values, choices, split) replaying the reference's four random draws; with
    sampler="neighbor" (reference :92-93) the edge draw is dict(u_vertex, tries) instead of `edges` (sample_edge_neighborhood)
    and `adj` may carry a prebuilt AdjIndex of `triplets` (the reference builds its adjacency lists once per run)."""
    if sampler not in ("uniform", "neighbor"):
        raise ValueError("Sampler type must be either 'uniform' or 'neighbor'.")
    draws = draws or {}
    dev = triplets.device
    T = int(triplets.shape[0])
    pick = draws.get("edges")
    if pick is not None:
        pick = torch.as_tensor(pick).to(dev).long()
    elif sampler == "uniform":
        pick = sample_edge_uniform(T, sample_size, dev, generator)
    else:
        adj = adj if adj is not None else AdjIndex(num_nodes, triplets)
        pick = sample_edge_neighborhood(adj, sample_size, {k: draws[k] for k in ("u_vertex", "tries") if k in draws} or None, generator)
    edges = triplets[pick].long()
    uniq_v, src, dst = relabel_nodes(edges[:, 0], edges[:, 2], num_nodes)
    rel = edges[:, 1].contiguous()
    relabeled = torch.stack((src, rel, dst), dim=1)
    samples, labels = negative_sampling(relabeled, int(uniq_v.numel()), negative_rate, draws.get("values"), draws.get("choices"), generator)
    n_split = int(sample_size * split_size)
    split = draws.get("split")
    split = torch.randperm(int(sample_size), device=dev, generator=generator)[:n_split] if split is None else torch.as_tensor(split).to(dev).long()
    g = G.build_search_graph(int(uniq_v.numel()), num_rels, relabeled[split], device=dev)
    src_o, _, _ = g.edges(form="all")
    deg = g._in_degree32.long()
    node_norm = G._deg_norm_table(dev, int(g.num_edges()) + 1)[deg]
    return g, uniq_v, src_o, g.edata["e_type"], node_norm, samples, labels


class LabelIndex:
    """``sr2o`` of process() (reference utils/process_data.py:4-31) resident in HBM: for every (subject, relation) pair
    -- relations r and r + num_rel for the inverse direction -- the set of known objects, as ascending keys
    ``subject * 2R + relation`` with a CSR of object ids.  ``labels(subj, rel)`` is TrainDataset / TestDataset.get_label
    for a batch (reference utils/data_set.py:21-33), optionally label-smoothed like TrainDataset.__getitem__ (:21-23)."""

    def __init__(self, triples, num_rel, num_ent, device):
        t = torch.as_tensor(np.asarray(triples) if not torch.is_tensor(triples) else triples).to(device).long()
        s, r, o = t[:, 0], t[:, 1], t[:, 2]
        self.num_rel, self.num_ent = int(num_rel), int(num_ent)
        key = torch.cat((s * (2 * num_rel) + r, o * (2 * num_rel) + r + num_rel))
        obj = torch.cat((o, s))
        pair = torch.unique(key * num_ent + obj)                          # the sets of sr2o: duplicates collapse; ascending
        key_s, obj_s = pair // num_ent, pair % num_ent
        self.keys, counts = torch.unique_consecutive(key_s, return_counts=True)
        rowptr = torch.zeros(self.keys.numel() + 1, dtype=torch.int64, device=t.device)
        rowptr[1:] = torch.cumsum(counts, 0)
        self.rowptr, self.objs = rowptr.to(torch.int32).contiguous(), obj_s.to(torch.int32).contiguous()
        G.settle(t.device)

    def labels(self, subj, rel, label_smooth=0.0):
        subj, rel = subj.long(), rel.long()
        q = (subj * (2 * self.num_rel) + rel).contiguous()
        require_hip(q)
        B = int(q.numel())
        out = torch.empty(B, self.num_ent, dtype=torch.float32, device=q.device)
        v = torch.tensor([0.0, 1.0])
        if label_smooth != 0.0:                                             # the reference's own float32 expression (data_set.py:23)
            v = (1.0 - label_smooth) * v + (1.0 / self.num_ent)
        call("mrg_multi_hot_labels", (ptr(self.keys), ptr(self.rowptr), ptr(self.objs), ptr(q), B, int(self.keys.numel()), self.num_ent,
                                      float(v[0]), float(v[1]), ptr(out), stream_of(q)))
        return out
