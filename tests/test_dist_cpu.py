"""world_size-2 (and 3) gloo tests of the relation-block sharded supernet step: partitioning,
collectives with adjoint backward, SyncBatchNorm and the flat gradient all-reduce must
reproduce the single-process result of the reference (golden vectors)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden, sub
from mr_gnas_amd import dist as MD


def test_relation_block_cuts_balanced_and_group_aligned():
    rng = np.random.default_rng(0)
    E, R, N = 20000, 11, 500
    w = np.array([0.4, 0.35] + [0.25 / 9] * 9)              # two relations hold 3/4 of the edges (WN18RR-like)
    et = np.sort(rng.choice(R, size=E, p=w))
    dst = rng.integers(0, N, size=E)
    order = np.lexsort((dst, et))
    et, dst = et[order], dst[order]
    for parts in (2, 4, 8):
        cuts = MD.relation_block_cuts(et, dst, parts)
        assert cuts[0] == 0 and cuts[-1] == E and all(b >= a for a, b in zip(cuts, cuts[1:]))
        sizes = np.diff(cuts)
        assert sizes.max() <= 1.15 * E / parts, (parts, sizes)
        for c in cuts[1:-1]:                                   # never inside a (relation, dst) group
            assert (et[c - 1], dst[c - 1]) != (et[c], dst[c])
    assert MD.node_ranges(10, 4) == [0, 3, 6, 8, 10]
    assert MD.relation_block_cuts(np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64), 3) == [0, 0, 0, 0]


def test_world8_partition_of_the_wn18rr_shape():
    """BASELINE config 3's skew: 11 relations (22 directed ids), the two largest hold about half of the triples.  An
    8-way relation-block partition must stay balanced (relations bigger than E / 8 are split by dst range), never cut
    a (relation, dst) group, keep [in-edges | out-edges] contiguous per shard, and cover every edge exactly once."""
    from mr_gnas_amd import graph as G, synth
    n, r, t = synth.SHAPES["wn18rr"]
    tri = synth.synth_kg(n, r, t, 0)
    g = G.build_search_graph(n, r, tri)
    src, dst, _ = g.edges(form="all")
    et = g.edata["e_type"]
    hist = np.bincount(et.numpy(), minlength=2 * r)
    assert hist.max() > g.num_edges() / 8                        # at least one relation alone exceeds a rank's share
    seen = np.zeros(g.num_edges(), dtype=np.int64)
    sizes = []
    for rank in range(8):
        sh = MD.EdgeShard(n, src, dst, et, g.edata["norm"], r, rank, 8, "cpu")
        ids = sh.global_edge_ids.numpy()
        seen[ids] += 1
        sizes.append(len(ids))
        e = sh.edata["e_type"].numpy()
        b0, b1 = sh.bounds()
        assert (e[:b0] < r).all() and (e[b0:] >= r).all() and b1 == len(ids)     # [original | inverse] halves stay contiguous
        assert sh.node_hi - sh.node_lo <= sh.node_chunk and sh.node_cuts[-1] == n
    assert (seen == 1).all()
    assert max(sizes) <= 1.1 * g.num_edges() / 8, sizes
    # the tensor formulation of the same partition (what EdgeShard runs for device tensors: one stable sort by relation * n + dst)
    for rank in (0, 3, 7):
        host = MD.EdgeShard(n, src, dst, et, g.edata["norm"], r, rank, 8, "cpu")
        dev = MD.EdgeShard.__new__(MD.EdgeShard)
        dev._init_on_device(n, src, dst, et, g.edata["norm"], r, rank, 8, "cpu")
        assert host.cuts == dev.cuts and host.bounds() == dev.bounds()
        assert torch.equal(host.global_edge_ids, dev.global_edge_ids)
        assert torch.equal(host.edata["norm"], dev.edata["norm"]) and torch.equal(host.global_in_degree, dev.global_in_degree)
    cuts, chunk = MD.node_chunks(n, 8)
    assert chunk * 8 >= n and cuts[0] == 0 and cuts[-1] == n and all(b - a in (chunk, n - 7 * chunk) for a, b in zip(cuts, cuts[1:]))
    assert MD.node_chunks(5, 8) == ([0, 1, 2, 3, 4, 5, 5, 5, 5], 1) and MD.node_chunks(0, 3) == ([0, 0, 0, 0], 0)


def _worker(rank, world, port, case, out):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_kernels as CK
    from mr_gnas_amd import supernet as S
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    try:
        z = load_golden(case)
        n = z["node_id"].numel()
        net = S.SearchNetwork("cpu", z["Nall"], z["R"], z["layers"], 1, 2, 2, z["D"], z["D0"], z["nbase"], 9.0, 0.0, 0.0,
                              registry=CK.registry())
        net.load_state_dict({**sub(z, "param/"), **sub(z, "buffer/")})
        net.load_alpha([z[f"alpha/{i}"] for i in range(5)])
        net.train()
        shard = MD.EdgeShard(n, z["src"], z["dst"], z["edge_type"], z["norm"], z["R"], rank, world, "cpu")
        sn = MD.ShardedSupernet(net, shard, z["node_id"], kernels=CK)
        ent, rel = sn.forward()
        lo = MD.node_ranges(len(z["data"]), world)
        loss = sn.loss(ent, rel, z["data"][lo[rank]:lo[rank + 1]], z["labels"][lo[rank]:lo[rank + 1]], len(z["data"]))
        loss.backward()
        params = list(net.parameters())
        MD.all_reduce_gradients(params + net.arch_parameters()[:4])
        total = loss.detach().clone()
        dist.all_reduce(total)
        if rank == 0:
            res = {"ent": ent.detach(), "rel": rel.detach(), "loss": total,
                   "edges": [int(shard.num_edges())]}
            for k, p in net.named_parameters():
                res["g/" + k] = p.grad
            for i in range(4):
                res[f"ga/{i}"] = net.arch_parameters()[i].grad
            torch.save(res, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, "supernet_tiny"), (2, "supernet_d24"), (3, "supernet_tiny")])
def test_sharded_supernet_matches_reference(tmp_path, world, case):
    out = str(tmp_path / "res.pt")
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, case, out), nprocs=world, join=True)
    res = torch.load(out)
    z = load_golden(case)
    torch.testing.assert_close(res["ent"], z["ent"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(res["rel"], z["rel_out"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(res["loss"], z["loss"], rtol=1e-5, atol=1e-6)
    for i in range(4):
        ref = z[f"galpha/{i}"]
        assert float((res[f"ga/{i}"] - ref).abs().max()) <= 2e-3 * max(float(ref.abs().max()), 1e-8) + 1e-7, f"alpha {i}"
    for k, v in sub(z, "gparam/").items():
        got = res["g/" + k]
        scale = max(float(v.abs().max()), 1e-6)
        assert float((got - v).abs().max()) <= 2e-3 * scale + 5e-6, k


README_CELL = [('pre_sub', 1, 0), ('f_sparse_comp', 2, 1), ('f_sparse_comp', 3, 2), ('a_max', 4, 2), ('a_max', 5, 3),
               ('f_sparse_last', 6, 5), ('f_sparse_last', 7, 5)]


def _fixed_worker(rank, world, port, case, out):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_kernels as CK
    from mr_gnas_amd import supernet as S
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    try:
        z = load_golden(case)
        geno = [S.Genotype(alpha_cell=README_CELL, concat_node=[4, 5, 6, 7], score_func='sf_DisMult')]
        net = S.FixedNetwork("cpu", geno, z["N"], z["R"], z["D"], z["D0"], z["nbase"], registry=CK.registry())
        net.load_state_dict({**sub(z, "param/"), **sub(z, "buffer/")})
        net.train()
        shard = MD.EdgeShard(z["N"], z["src"], z["dst"], z["etype"], z["norm"], z["R"], rank, world, "cpu")
        sn = MD.ShardedFixedNet(net, shard, kernels=CK)
        pred = sn.forward(z["subj"], z["rel"])
        loss = sn.loss(pred, z["label"])
        loss.backward()
        MD.all_reduce_gradients(sn.replicated_parameters())
        total = loss.detach().clone()
        dist.all_reduce(total)
        # the row-sharded pieces travel to rank 0 for the comparison only: prediction columns and the own rows' table gradient
        preds = [None] * world
        gembs = [None] * world
        dist.all_gather_object(preds, pred.detach())
        dist.all_gather_object(gembs, sn.emb_own.grad)
        if rank == 0:
            res = {"pred": torch.cat(preds, dim=1), "loss": total, "g/embedding_h.weight": torch.cat(gembs, dim=0),
                   "own": [int(p.shape[1]) for p in preds], "replicated_has_table": any(p is net.embedding_h.weight for p in sn.replicated_parameters())}
            for k, p in net.named_parameters():
                if k != "embedding_h.weight":
                    res["g/" + k] = p.grad
            torch.save(res, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, "fixednet_d64"), (2, "fixednet_tiny"), (3, "fixednet_tiny")])
def test_sharded_fixed_network_matches_reference(tmp_path, world, case):
    """The fixed-genotype network (reference models/model_lp.py:77-150, README genotype) on relation blocks with ROW-SHARDED node
    tables (dist.ShardedFixedNet) against the reference's own prediction, loss and gradients."""
    out = str(tmp_path / "res.pt")
    port = 29500 + (os.getpid() % 2000) + 10 + world
    mp.spawn(_fixed_worker, args=(world, port, case, out), nprocs=world, join=True)
    res = torch.load(out)
    z = load_golden(case)
    assert sum(res["own"]) == z["N"] and not res["replicated_has_table"]
    torch.testing.assert_close(res["pred"], z["pred"], rtol=1e-4, atol=2e-6)
    torch.testing.assert_close(res["loss"], z["loss"], rtol=1e-5, atol=1e-7)
    for k, v in sub(z, "gparam/").items():
        got = res["g/" + k]
        assert got is not None, k
        scale = max(float(v.abs().max()), 1e-8)
        assert float((got - v).abs().max()) <= 2e-3 * scale + 1e-7, k


MIXED_CELL = [('pre_mult', 1, 0), ('f_identity', 2, 1), ('f_comp', 3, 1), ('f_zero', 3, 2), ('a_sum', 4, 2), ('a_mean', 5, 3),
              ('f_identity', 6, 4), ('f_dense_last', 7, 5), ('f_zero', 7, 4)]


def _kinds_worker(rank, world, port, out):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_kernels as CK
    from mr_gnas_amd import supernet as S
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    try:
        z = load_golden("fixednet_d64")
        geno = [S.Genotype(alpha_cell=MIXED_CELL, concat_node=[4, 5, 6, 7], score_func='sf_DisMult')] * 2      # two cells: the all-gather between them
        torch.manual_seed(3)
        net = S.FixedNetwork("cpu", geno, z["N"], z["R"], z["D"], z["D0"], z["nbase"], registry=CK.registry())
        net.train()
        shard = MD.EdgeShard(z["N"], z["src"], z["dst"], z["etype"], z["norm"], z["R"], rank, world, "cpu")
        sn = MD.ShardedFixedNet(net, shard, kernels=CK)
        kinds = [sn._node_rows(m) for cell in net.cells[:1] for n in range(cell._nb) for i in range(n + 1) for m in cell._ops[n][i]]
        pred = sn.forward(z["subj"], z["rel"])
        loss = sn.loss(pred, z["label"])
        loss.backward()
        MD.all_reduce_gradients(sn.replicated_parameters())
        total = loss.detach().clone()
        dist.all_reduce(total)
        preds, gembs = [None] * world, [None] * world
        dist.all_gather_object(preds, pred.detach())
        dist.all_gather_object(gembs, sn.emb_own.grad)
        if rank == 0:
            torch.save({"pred": torch.cat(preds, dim=1), "loss": total, "kinds": kinds, "gemb": torch.cat(gembs, dim=0),
                        "g": {k: p.grad for k, p in net.named_parameters() if k != "embedding_h.weight"}}, out)
    finally:
        dist.destroy_process_group()


def test_sharded_fixed_network_other_genotypes_world2_equals_world1(tmp_path):
    """A genotype with every row kind the README one lacks -- pre_mult (no BatchNorm), f_identity / f_zero in the first AND the last
    stage (their row kind comes from their siblings or their input: ShardedFixedNet._node_rows), a_sum and a_mean exchanges, two ops
    summed into one node, two cells (the all-gather between them): two ranks must reproduce one rank."""
    res = {}
    for world in (1, 2):
        out = str(tmp_path / f"res{world}.pt")
        port = 29500 + (os.getpid() % 2000) + 20 + world
        mp.spawn(_kinds_worker, args=(world, port, out), nprocs=world, join=True)
        res[world] = torch.load(out)
    # zero node, node 2, node 3 (two ops): edge rows; nodes 4-7 (five ops): node rows
    assert res[1]["kinds"] == [False, False, False, False, True, True, True, True, True]
    torch.testing.assert_close(res[2]["pred"], res[1]["pred"], rtol=1e-4, atol=1e-6)          # float32, two cells, other summation orders
    torch.testing.assert_close(res[2]["loss"], res[1]["loss"], rtol=1e-6, atol=1e-8)
    assert float((res[2]["gemb"] - res[1]["gemb"]).abs().max()) <= 1e-4 * float(res[1]["gemb"].abs().max()) + 1e-9
    for k, v in res[1]["g"].items():
        got = res[2]["g"][k]
        assert (got is None) == (v is None), k
        if v is not None:
            assert float((got - v).abs().max()) <= 1e-4 * float(v.abs().max()) + 1e-6, k    # (+ 1e-6: gradients that are zero in exact arithmetic -- a scale in front of a BatchNorm -- are 1e-7 of noise on both sides)


def _bring_up_worker(rank, world, port, fault, out):
    """rccl.bring_up with a fault injected on ONE rank: every rank must leave it together with None (advisor r4)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mr_gnas_amd import rccl

        class FakeLib:
            def ncclGetUniqueId(self, ref):
                return 1 if fault == "unique_id" else 0

            def ncclGetErrorString(self, code):
                return b"injected"

        def load():
            if fault == "load" and rank == 1:
                raise rccl.RcclError("injected: no librccl.so on this rank")
            return FakeLib()

        class FakeComm:
            def __init__(self, r, w, d, unique_id=None):
                if fault == "init" and r == 1:
                    raise rccl.RcclError("injected: ncclCommInitRank failed on this rank")
                self.aborted = False

            def all_reduce(self, t, op):
                pass

            def abort(self):
                self.aborted = True

        rccl.load, rccl.Comm = load, FakeComm
        got = rccl.bring_up(rank, world, "cpu", timeout_s=20.0)
        torch.save({"none": got is None}, f"{out}.{rank}")
        dist.barrier()                          # the ranks are still in step: a mismatched collective would hang or raise here
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fault", ["load", "unique_id", "init"])
def test_direct_rccl_bring_up_fails_on_every_rank_together(tmp_path, fault):
    out = str(tmp_path / "bring")
    port = 31500 + (os.getpid() % 2000) + len(fault)
    mp.spawn(_bring_up_worker, args=(2, port, fault, out), nprocs=2, join=True)
    assert all(torch.load(f"{out}.{r}")["none"] for r in range(2))
