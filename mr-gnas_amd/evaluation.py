"""Filtered ranking of the fixed-genotype driver on the device (reference train/mr_lp_train.py:269-358).

``filtered_ranks`` is the body of the reference's ``predict`` loop (:290-299) as one HIP kernel
(``mrg_rank_filtered``): instead of a masked copy of the [B, N] score matrix and two full argsorts per batch, one
pass counts, per row, the entries that beat the target.  Ties follow a stable descending sort (the reference's
``torch.argsort`` leaves the order of equal scores unspecified).  ``predict`` / ``combine_results`` mirror the
reference's functions of the same role so a driver can swap them in.
"""
import torch
import torch.nn.functional as F

from ._lib import call, ptr, require_hip, stream_of, f32c


def filtered_ranks(pred, labels, obj):
    """ranks [B] int64 (1-based): position of obj[b] among the entities whose label is zero (plus itself)."""
    pred, labels, obj = f32c(pred), f32c(labels), obj.long().contiguous()
    require_hip(pred, labels, obj)
    B, N = pred.shape
    if labels.shape != pred.shape or obj.numel() != B:
        raise ValueError("filtered_ranks: pred / labels [B, N] and obj [B] expected")
    ranks = torch.empty(B, dtype=torch.int64, device=pred.device)
    call("mrg_rank_filtered", (ptr(pred), ptr(labels), ptr(obj), B, N, ptr(ranks), stream_of(pred)), nbytes=8 * B * N)
    return ranks


_HITS = (1, 3, 10)


def predict(val_test_loader, g, model, device):
    """The evaluation pass of the fixed-genotype driver (reference train/mr_lp_train.py:269-314) with its bookkeeping on the device:
    the per-batch rank statistics (number of predictions, sum of ranks, sum of reciprocal ranks, hits at 1 / 3 / 10) and the summed
    BCE loss are accumulated in ONE float64 device vector and read back by ONE copy when the loader is exhausted -- the reference
    synchronises three times per batch (`.item()`).  Returns (results, summed loss), results keyed 'count', 'mr', 'mrr', 'hits@k'
    like the reference's, counts as int."""
    ks = torch.tensor(_HITS, dtype=torch.float64, device=device)
    acc = torch.zeros(4 + len(_HITS), dtype=torch.float64, device=device)       # [count, mr, mrr, loss, hits@k ...]
    with torch.no_grad():
        model.eval()
        for triplets, labels in val_test_loader:
            triplets, labels = triplets.to(device), labels.to(device)
            pred = model(g, triplets[:, 0], triplets[:, 1])
            ranks = filtered_ranks(pred, labels, triplets[:, 2]).double()
            batch = torch.cat((torch.stack((ranks.new_tensor(float(ranks.numel())), ranks.sum(), ranks.reciprocal().sum(),
                                            F.binary_cross_entropy(pred, labels).double())),
                               (ranks.unsqueeze(0) <= ks.unsqueeze(1)).sum(dim=1).double()))
            acc += batch
    tot = acc.tolist()                                                          # the pass's only device -> host copy
    results = {'count': int(tot[0]), 'mr': tot[1], 'mrr': tot[2]}
    results.update({f'hits@{k}': int(tot[4 + i]) for i, k in enumerate(_HITS)})
    return results, tot[3]


def combine_results(left, right):
    """get_combined_results of the reference's infer() (train/mr_lp_train.py:327-345)."""
    if left['count'] != right['count']:
        raise AssertionError("head and tail evaluations must cover the same number of triples")
    count, results = float(left['count']), {}
    for side, r in (('left', left), ('right', right)):
        results[f'{side}_mr'] = round(r['mr'] / count, 5)
        results[f'{side}_mrr'] = round(r['mrr'] / count, 5)
    results['mr'] = round((left['mr'] + right['mr']) / (2 * count), 5)
    results['mrr'] = round((left['mrr'] + right['mrr']) / (2 * count), 5)
    for k in [1, 3, 10]:
        results[f'left_hits@{k}'] = round(left[f'hits@{k}'] / count, 5)
        results[f'right_hits@{k}'] = round(right[f'hits@{k}'] / count, 5)
        results[f'hits@{k}'] = round((results[f'left_hits@{k}'] + results[f'right_hits@{k}']) / 2, 5)
    return results
