"""Shape-matched synthetic knowledge graphs (no dataset files, no network).

The reference downloads FB15k-237 / WN18RR through DGL; here every workload is
a seeded synthetic KG with the public datasets' sizes: node popularity ~
rank^-0.75 (ids randomly permuted), relation frequency ~ rank^-1.
"""
import numpy as np

SHAPES = {
    # name: (num_nodes, num_rels, num_train_triples)
    "fb15k237": (14541, 237, 272115),
    "wn18rr": (40943, 11, 86835),
    "synthetic10m": (1_000_000, 256, 5_000_000),
}


def synth_kg(num_nodes, num_rels, num_triples, seed=0):
    """[T, 3] int64 (s, r, o) triples."""
    rng = np.random.default_rng(seed)
    pn = 1.0 / np.arange(1, num_nodes + 1) ** 0.75
    pn /= pn.sum()
    ids = rng.permutation(num_nodes)
    pr = 1.0 / np.arange(1, num_rels + 1)
    pr /= pr.sum()
    s = ids[rng.choice(num_nodes, size=num_triples, p=pn)]
    o = ids[rng.choice(num_nodes, size=num_triples, p=pn)]
    r = rng.choice(num_rels, size=num_triples, p=pr)
    return np.stack([s, r, o], axis=1).astype(np.int64)


def negative_sampling(pos, num_nodes, rate, rng):
    """Corrupt subject or object uniformly at random, `rate` negatives per positive
    (the scheme of reference utils/utils_rgcn.py:191-204).  Returns (samples [T*(rate+1), 3], labels)."""
    T = len(pos)
    neg = np.tile(pos, (rate, 1))
    labels = np.zeros(T * (rate + 1), dtype=np.float32)
    labels[:T] = 1
    values = rng.integers(0, num_nodes, size=T * rate)
    subj = rng.uniform(size=T * rate) > 0.5
    neg[subj, 0] = values[subj]
    neg[~subj, 2] = values[~subj]
    return np.concatenate((pos, neg)), labels


def sample_step_graph(triples, sample_size, split=0.5, negative_rate=10, seed=0):
    """One search-step sample (the scheme of reference utils/utils_rgcn.py:79-118): draw
    `sample_size` triples uniformly, relabel nodes, build negatives, keep `split` of the
    positives as graph structure.  Returns (node_id, graph_triples, samples, labels)."""
    rng = np.random.default_rng(seed)
    pick = rng.choice(len(triples), size=min(sample_size, len(triples)), replace=False)
    s, r, o = triples[pick].T
    node_id, inv = np.unique(np.concatenate((s, o)), return_inverse=True)
    s2, o2 = inv[: len(s)], inv[len(s):]
    rel = np.stack((s2, r, o2), axis=1)
    samples, labels = negative_sampling(rel, len(node_id), negative_rate, rng)
    keep = rng.choice(len(rel), size=int(len(rel) * split), replace=False)
    return node_id, rel[keep], samples, labels
