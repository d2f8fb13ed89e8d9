"""Oracle restatement (CPU numpy / torch, test-only; see oracle/__init__.py) of the data preparation either side of the
hot path: negative sampling, node relabelling and the sampled step graph (reference utils/utils_rgcn.py:79-118, 191-204),
the sr2o label sets and dense targets (reference utils/process_data.py:4-31, utils/data_set.py:15-33) and the filtered
ranking of predict() (reference train/mr_lp_train.py:290-299).  Random draws are explicit arguments.  Pinned by
tests/golden/sampling_small.npz, sampling_neighbor_*.npz and labels_ranking_small.npz, which were produced by running the
reference itself."""
import collections

import numpy as np
import torch

from .graph import build_search_graph


def negative_sampling(pos, num_entity, rate, values, choices):
    # reference utils/utils_rgcn.py:191-204
    pos = np.asarray(pos)
    B = len(pos)
    neg = np.tile(pos, (rate, 1))
    labels = np.zeros(B * (rate + 1), dtype=np.float32)
    labels[:B] = 1
    values, choices = np.asarray(values), np.asarray(choices)
    subj, obj = choices > 0.5, choices <= 0.5
    neg[subj, 0] = values[subj]
    neg[obj, 2] = values[obj]
    return np.concatenate((pos, neg)), labels


def adjacency(num_nodes, triplets):
    # reference utils/utils_rgcn.py:18-28 (get_adj_and_degrees): per vertex the [triple id, other end] pairs in append order
    adj = [[] for _ in range(num_nodes)]
    for i, (s, _, o) in enumerate(np.asarray(triplets)):
        adj[s].append((i, o))
        adj[o].append((i, s))
    return [np.asarray(a, dtype=np.int64).reshape(-1, 2) for a in adj], np.array([len(a) for a in adj])


def sample_edge_neighborhood(adj, degrees, n_triplets, sample_size, u_vertex, tries):
    """Reference utils/utils_rgcn.py:30-71 with its draws given: u_vertex[i] is the uniform behind the i-th
    np.random.choice(n, p=probabilities) (legacy choice: searchsorted of the normalised cumsum, side='right'); `tries` the
    adjacency slots its np.random.choice(np.arange(m)) calls returned, rejected ones included."""
    counts = np.array([d for d in degrees])
    picked = np.zeros(n_triplets, dtype=bool)
    seen = np.zeros(len(degrees), dtype=bool)
    edges = np.zeros(sample_size, dtype=np.int32)
    t = 0
    for i in range(sample_size):
        w = counts * seen
        if np.sum(w) == 0:
            w = np.ones_like(w)
            w[np.where(counts == 0)] = 0
        p = w / np.sum(w)
        cdf = p.cumsum()
        cdf /= cdf[-1]
        v = int(cdf.searchsorted(u_vertex[i], side="right"))
        seen[v] = True
        e, o = adj[v][tries[t]]
        t += 1
        while picked[e]:
            e, o = adj[v][tries[t]]
            t += 1
        edges[i] = e
        picked[e] = True
        counts[v] -= 1
        counts[o] -= 1
        seen[o] = True
    return edges


def sampled_graph_and_labels(triplets, sample_size, split_size, num_rels, negative_rate, draws):
    """Reference utils/utils_rgcn.py:79-118 with its four draws given: edges, values, choices, split."""
    e = np.asarray(triplets)[np.asarray(draws["edges"])]
    src, rel, dst = e.transpose()
    uniq_v, inv = np.unique((src, dst), return_inverse=True)
    src, dst = np.reshape(inv, (2, -1))
    relabeled = np.stack((src, rel, dst)).transpose()
    samples, labels = negative_sampling(relabeled, len(uniq_v), negative_rate, draws["values"], draws["choices"])
    ids = np.asarray(draws["split"])[: int(sample_size * split_size)]
    g = build_search_graph(len(uniq_v), num_rels, relabeled[ids])
    deg = np.bincount(g.dst.numpy(), minlength=len(uniq_v))
    with np.errstate(divide="ignore"):
        node_norm = deg.astype(np.float32) ** np.float32(-0.5)
    node_norm[np.isinf(node_norm)] = 0
    return g, uniq_v, g.src.numpy(), g.etype.numpy(), node_norm, samples, labels


def sr2o(triples, num_rel):
    # reference utils/process_data.py:11-20
    d = collections.defaultdict(set)
    for s, r, o in np.asarray(triples).tolist():
        d[(s, r)].add(o)
        d[(o, r + num_rel)].add(s)
    return d


def dense_labels(index, subj, rel, num_ent, label_smooth=0.0):
    # reference utils/data_set.py:15-33
    out = np.zeros((len(subj), num_ent), dtype=np.float32)
    for i, (s, r) in enumerate(zip(np.asarray(subj).tolist(), np.asarray(rel).tolist())):
        out[i, list(index.get((s, r), ()))] = 1
    y = torch.from_numpy(out)
    if label_smooth != 0.0:
        y = (1.0 - label_smooth) * y + (1.0 / num_ent)
    return y


def filtered_ranks(pred, labels, obj):
    # reference train/mr_lp_train.py:290-299 (stable sort: equal scores keep their index order)
    pred, labels, obj = torch.as_tensor(pred).clone(), torch.as_tensor(labels), torch.as_tensor(obj).long()
    b = torch.arange(pred.shape[0])
    target = pred[b, obj]
    pred = torch.where(labels.to(torch.uint8).bool(), -torch.ones_like(pred) * 10000000, pred)
    pred[b, obj] = target
    order = torch.sort(pred, dim=1, descending=True, stable=True).indices
    return 1 + torch.argsort(order, dim=1)[b, obj]
