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


def predict(val_test_loader, g, model, device):
    """Reference train/mr_lp_train.py:269-314: returns (results, summed loss) with results['count'|'mr'|'mrr'|'hits@k']."""
    with torch.no_grad():
        results, test_loss = {}, []
        model.eval()
        for triplets, labels in val_test_loader:
            triplets, labels = triplets.to(device), labels.to(device)
            subj, rel, obj = triplets[:, 0], triplets[:, 1], triplets[:, 2]
            pred = model(g, subj, rel)
            test_loss.append(F.binary_cross_entropy(pred, labels).item())
            ranks = filtered_ranks(pred, labels, obj).float()
            results['count'] = torch.numel(ranks) + results.get('count', 0)
            results['mr'] = torch.sum(ranks).item() + results.get('mr', 0)
            results['mrr'] = torch.sum(1.0 / ranks).item() + results.get('mrr', 0)
            for k in [1, 3, 10]:
                results[f'hits@{k}'] = torch.numel(ranks[ranks <= k]) + results.get(f'hits@{k}', 0)
        return results, float(sum(test_loss))


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
