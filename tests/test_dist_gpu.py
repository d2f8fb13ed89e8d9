"""The sharded step on RCCL bound directly (mr_gnas_amd/rccl.py): communicator of one rank on the GPU box -- the collectives are
real RCCL launches on the step's own HIP stream -- against the reference golden, eager and replayed from a HIP graph; the
device-side relation-block partition against the host one.  Multi-rank arithmetic of the same code: tests/test_dist_cpu.py (gloo)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from mr_gnas_amd import dist as MD, supernet as S
from test_nets_gpu import grads_close, load_net_state

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def comm():
    from mr_gnas_amd import rccl
    c = rccl.Comm(0, 1, DEV)
    yield c
    c.destroy()


def _golden_net(case):
    z = load_golden(case)
    net = S.SearchNetwork(DEV, z["Nall"], z["R"], z["layers"], 1, 2, 2, z["D"], z["D0"], z["nbase"], 9.0, 0.0, 0.0).to(DEV)
    load_net_state(net, z)
    net.load_alpha([z[f"alpha/{i}"].to(DEV) for i in range(5)])
    net.train()
    return z, net


def test_direct_rccl_collectives_of_one_rank(comm):
    x = torch.randn(1000, 64, device=DEV)
    y = x.clone()
    comm.all_reduce(y, "sum")
    out = torch.empty_like(x)
    comm.reduce_scatter_tensor(out, x, "max")
    full = torch.empty_like(x)
    comm.all_gather_into_tensor(full, x)
    torch.cuda.synchronize()
    assert torch.equal(y, x) and torch.equal(out, x) and torch.equal(full, x)
    with pytest.raises(Exception):
        comm.reduce_scatter_tensor(torch.empty(10, device=DEV), torch.empty(30, device=DEV), "sum")


@pytest.mark.parametrize("case", ["supernet_d24", "supernet_d200_sampled"])
def test_sharded_step_on_direct_rccl_matches_reference(case, comm):
    z, net = _golden_net(case)
    n = z["node_id"].numel()
    shard = MD.EdgeShard(n, z["src"].to(DEV), z["dst"].to(DEV), z["edge_type"].to(DEV), z["norm"].to(DEV), z["R"], 0, 1, DEV)
    sn = MD.ShardedSupernet(net, shard, z["node_id"], group=comm)
    before = comm.launches
    ent, rel = sn.forward()
    loss = sn.loss(ent, rel, z["data"].to(DEV), z["labels"].to(DEV), len(z["data"]))
    loss.backward()
    MD.all_reduce_gradients(list(net.parameters()) + net.arch_parameters()[:4], comm)
    assert comm.launches - before >= 20                    # exchanges, statistics, layer all-gathers, the flat gradient all-reduce
    torch.testing.assert_close(ent.cpu(), z["ent"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(loss.detach().cpu(), z["loss"], rtol=1e-4, atol=1e-6)
    grads_close(net, z, 2e-3, "sharded, direct RCCL")


def test_sharded_step_replays_from_a_hip_graph(comm):
    """Forward + loss + backward + flat gradient all-reduce of the sharded step captured ONCE and replayed: the RCCL launches sit in
    the graph like kernels (no c10d work objects, no watchdog).  The replayed loss and gradients are those of the eager step."""
    z, net = _golden_net("supernet_d200_sampled")
    n = z["node_id"].numel()
    shard = MD.EdgeShard(n, z["src"].to(DEV), z["dst"].to(DEV), z["edge_type"].to(DEV), z["norm"].to(DEV), z["R"], 0, 1, DEV)
    data, labels = z["data"].to(DEV), z["labels"].to(DEV)
    params = list(net.parameters()) + net.arch_parameters()[:4]
    sn = MD.ShardedSupernet(net, shard, z["node_id"], group=comm)
    static = {}

    def step():
        for p in params:
            p.grad = None
        ent, rel = sn.forward()
        loss = sn.loss(ent, rel, data, labels, len(data))
        loss.backward()
        MD.all_reduce_gradients(params, comm)
        static["loss"], static["ent"] = loss.detach(), ent.detach()
        static["grads"] = [p.grad for p in params]

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()                                          # plans, workspaces, allocator pools
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    eager_loss = float(static["loss"])
    eager_grads = [g.clone() for g in static["grads"]]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        step()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert abs(float(static["loss"]) - eager_loss) <= 1e-6 * max(1.0, abs(eager_loss))
    torch.testing.assert_close(static["ent"].cpu(), z["ent"], rtol=1e-4, atol=2e-5)
    for a, b in zip(static["grads"], eager_grads):
        assert float((a - b).abs().max()) <= 1e-6 * max(float(b.abs().max()), 1e-6) + 1e-9


def test_device_partition_equals_the_host_partition():
    """EdgeShard built from device tensors (stable device sort by relation * n + dst) cuts and orders exactly like the host
    np.lexsort path: same cuts, same local edges in the same order, for every rank of a world of 8 -- on a graph whose largest
    relation exceeds a rank's share (the WN18RR situation, reference utils/utils_rgcn.py:151's order)."""
    rng = np.random.default_rng(5)
    n, R, E = 3000, 6, 40000
    et = rng.choice(2 * R, size=E, p=np.array([0.45, 0.25, 0.05, 0.05, 0.02, 0.02, 0.05, 0.05, 0.02, 0.02, 0.01, 0.01]))
    src, dst = rng.integers(0, n, E), rng.integers(0, n, E)
    norm = rng.random(E).astype(np.float32)
    for rank in range(8):
        host = MD.EdgeShard(n, src, dst, et, norm, R, rank, 8, DEV)
        dev = MD.EdgeShard(n, torch.from_numpy(src).to(DEV), torch.from_numpy(dst).to(DEV), torch.from_numpy(et).to(DEV),
                           torch.from_numpy(norm).to(DEV), R, rank, 8, DEV)
        assert host.cuts == dev.cuts and host.bounds() == dev.bounds()
        assert torch.equal(host.global_edge_ids, dev.global_edge_ids)
        for a, b in zip(host.edges(form="all")[:2], dev.edges(form="all")[:2]):
            assert torch.equal(a, b)
        assert torch.equal(host.edata["e_type"], dev.edata["e_type"]) and torch.equal(host.edata["norm"], dev.edata["norm"])
        assert torch.equal(host.global_in_degree, dev.global_in_degree)


def test_virtual_world_rehearsal_runs(comm):
    """rccl.VirtualWorld (bench.py --rehearse-shard): rank 3 of 8 on one GPU -- timing only, values finite."""
    from mr_gnas_amd import rccl
    z, net = _golden_net("supernet_d24")
    n = z["node_id"].numel()
    vw = rccl.VirtualWorld(3, 8, DEV)
    try:
        shard = MD.EdgeShard(n, z["src"].to(DEV), z["dst"].to(DEV), z["edge_type"].to(DEV), z["norm"].to(DEV), z["R"], 3, 8, DEV)
        sn = MD.ShardedSupernet(net, shard, z["node_id"], group=vw)
        ent, rel = sn.forward()
        lo = MD.node_ranges(len(z["data"]), 8)
        loss = sn.loss(ent, rel, z["data"][lo[3]:lo[4]].to(DEV), z["labels"][lo[3]:lo[4]].to(DEV), len(z["data"]))
        loss.backward()
        MD.all_reduce_gradients(list(net.parameters()) + net.arch_parameters()[:4], vw)
        torch.cuda.synchronize()
        assert torch.isfinite(loss) and ent.shape[0] == n and vw.launches >= 20
    finally:
        vw.destroy()


# ---------------------------------------------------------------------------------------------------------------------------
# the sharded FIXED-genotype network (reference models/model_lp.py:77-150; BASELINE C5 at N > 1): node tables row-sharded
# ---------------------------------------------------------------------------------------------------------------------------
def _fixed_golden(case):
    from test_nets_gpu import README_GENOTYPE
    from conftest import sub
    z = load_golden(case)
    net = S.FixedNetwork(DEV, README_GENOTYPE, z["N"], z["R"], z["D"], z["D0"], z["nbase"], score_args={"gamma": 9.0}).to(DEV)
    net.load_state_dict({**sub(z, "param/"), **sub(z, "buffer/")})
    net.train()
    return z, net


@pytest.mark.parametrize("case", ["fixednet_tiny", "fixednet_d64"])
def test_sharded_fixed_network_on_direct_rccl_matches_reference(case, comm):
    """dist.ShardedFixedNet on the HIP kernels with an RCCL communicator of one rank against the reference's own run: prediction,
    loss, every gradient (the row-sharded initial table's through its own-rows leaf).  Multi-rank arithmetic: the gloo tests."""
    from conftest import assert_param_grad
    z, net = _fixed_golden(case)
    shard = MD.EdgeShard(z["N"], z["src"].to(DEV), z["dst"].to(DEV), z["etype"].to(DEV), z["norm"].to(DEV), z["R"], 0, 1, DEV)
    sn = MD.ShardedFixedNet(net, shard, group=comm)
    before = comm.launches
    pred = sn.forward(z["subj"].to(DEV), z["rel"].to(DEV))
    loss = sn.loss(pred, z["label"].to(DEV))
    loss.backward()
    MD.all_reduce_gradients(sn.replicated_parameters(), comm)
    assert comm.launches - before >= 20                    # statistics, two aggregator exchanges, table all-gather, subject rows, gradients
    torch.testing.assert_close(pred.cpu(), z["pred"], rtol=1e-4, atol=5e-5)
    torch.testing.assert_close(loss.detach().cpu(), z["loss"], rtol=1e-4, atol=1e-6)
    assert net.embedding_h.weight.grad is None and not any(p is net.embedding_h.weight for p in sn.replicated_parameters())
    for k, p in net.named_parameters():
        g = sn.emb_own.grad if k == "embedding_h.weight" else p.grad
        assert_param_grad(z, k, g if g is not None else torch.zeros_like(p), 5e-4, 5e-6, case + " sharded")


def test_sharded_fixed_step_replays_from_a_hip_graph(comm):
    """Forward + loss + backward + gradient all-reduce + Adam of the sharded fixed-genotype step captured once and replayed."""
    z, net = _fixed_golden("fixednet_d64")
    shard = MD.EdgeShard(z["N"], z["src"].to(DEV), z["dst"].to(DEV), z["etype"].to(DEV), z["norm"].to(DEV), z["R"], 0, 1, DEV)
    sn = MD.ShardedFixedNet(net, shard, group=comm)
    subj, rel, label = z["subj"].to(DEV), z["rel"].to(DEV), z["label"].to(DEV)
    static = {}

    def step():
        for p in sn.parameters():
            p.grad = None
        pred = sn.forward(subj, rel)
        loss = sn.loss(pred, label)
        loss.backward()
        MD.all_reduce_gradients(sn.replicated_parameters(), comm)
        static["loss"], static["pred"] = loss.detach(), pred.detach()
        static["grads"] = [p.grad for p in sn.parameters()]

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    eager_loss, eager_grads = float(static["loss"]), [g.clone() for g in static["grads"]]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        step()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert abs(float(static["loss"]) - eager_loss) <= 1e-6 * max(1.0, abs(eager_loss))
    torch.testing.assert_close(static["pred"].cpu(), z["pred"], rtol=1e-4, atol=5e-5)
    for a, b in zip(static["grads"], eager_grads):
        assert float((a - b).abs().max()) <= 1e-6 * max(float(b.abs().max()), 1e-6) + 1e-9


def test_virtual_world_rehearsal_of_the_fixed_step_runs():
    """bench.py --rehearse-shard R/8 --workload c5_fixed_cell in small: rank 5 of 8 of the fixed-genotype step on one GPU (timing
    only: the other ranks contribute zeros), values finite, own columns = the rank's node chunk."""
    from mr_gnas_amd import rccl
    z, net = _fixed_golden("fixednet_d64")
    vw = rccl.VirtualWorld(5, 8, DEV)
    try:
        shard = MD.EdgeShard(z["N"], z["src"].to(DEV), z["dst"].to(DEV), z["etype"].to(DEV), z["norm"].to(DEV), z["R"], 5, 8, DEV)
        sn = MD.ShardedFixedNet(net, shard, group=vw)
        pred = sn.forward(z["subj"].to(DEV), z["rel"].to(DEV))
        loss = sn.loss(pred, z["label"].to(DEV))
        loss.backward()
        MD.all_reduce_gradients(sn.replicated_parameters(), vw)
        torch.cuda.synchronize()
        assert pred.shape == (len(z["subj"]), shard.n_own) and bool(torch.isfinite(pred).all()) and bool(torch.isfinite(loss))
        assert all(bool(torch.isfinite(p.grad).all()) for p in sn.parameters() if p.grad is not None)
    finally:
        vw.destroy()


# ---------------------------------------------------------------------------------------------------------------------------
# FOUR ranks on the HIP kernels: one process per rank sharing this box's one GPU, collectives over gloo on device tensors.
# The multi-rank arithmetic of tests/test_dist_cpu.py (CPU stand-in kernels) and the HIP kernels of the one-rank RCCL tests meet
# here: every rank runs the product's kernels on its relation block and the results must be the reference's.
# ---------------------------------------------------------------------------------------------------------------------------
def _hip_rank(rank, world, port, case, out):
    import datetime
    import os
    import torch.distributed as dist
    from conftest import sub
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    try:
        torch.cuda.set_device(0)
        z = load_golden(case)
        res = {}
        if case.startswith("fixednet"):
            z, net = _fixed_golden(case)
            shard = MD.EdgeShard(z["N"], z["src"].to(DEV), z["dst"].to(DEV), z["etype"].to(DEV), z["norm"].to(DEV), z["R"], rank, world, DEV)
            sn = MD.ShardedFixedNet(net, shard)
            pred = sn.forward(z["subj"].to(DEV), z["rel"].to(DEV))
            loss = sn.loss(pred, z["label"].to(DEV))
            loss.backward()
            MD.all_reduce_gradients(sn.replicated_parameters())
            preds, gembs = [None] * world, [None] * world
            dist.all_gather_object(preds, pred.detach().cpu())
            dist.all_gather_object(gembs, sn.emb_own.grad.cpu())
            res["pred"] = torch.cat(preds, dim=1)
            grads = {k: (torch.cat(gembs, dim=0) if k == "embedding_h.weight" else None if p.grad is None else p.grad.cpu())
                     for k, p in net.named_parameters()}
        else:
            z, net = _golden_net(case)
            n = z["node_id"].numel()
            shard = MD.EdgeShard(n, z["src"].to(DEV), z["dst"].to(DEV), z["edge_type"].to(DEV), z["norm"].to(DEV), z["R"], rank, world, DEV)
            sn = MD.ShardedSupernet(net, shard, z["node_id"])
            ent, rel = sn.forward()
            lo = MD.node_ranges(len(z["data"]), world)
            loss = sn.loss(ent, rel, z["data"][lo[rank]:lo[rank + 1]].to(DEV), z["labels"][lo[rank]:lo[rank + 1]].to(DEV), len(z["data"]))
            loss.backward()
            MD.all_reduce_gradients(list(net.parameters()) + net.arch_parameters()[:4])
            res["ent"] = ent.detach().cpu()
            grads = {k: None if p.grad is None else p.grad.cpu() for k, p in net.named_parameters()}
        total = loss.detach().clone()
        dist.all_reduce(total)
        torch.cuda.synchronize()
        if rank == 0:
            res.update(loss=total.cpu(), grads=grads, edges=int(shard.num_edges()))
            torch.save(res, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["supernet_d200_sampled", "fixednet_d64"])
def test_four_ranks_on_the_hip_kernels_match_reference(tmp_path, case):
    import os
    import time
    import torch.multiprocessing as mp
    from conftest import assert_param_grad
    world, out = 4, str(tmp_path / "res.pt")
    port = 29500 + (os.getpid() % 2000) + 40
    ctx = mp.start_processes(_hip_rank, args=(world, port, case, out), nprocs=world, join=False, start_method="spawn")
    deadline = time.time() + 420
    try:
        while not ctx.join(timeout=5):
            assert time.time() < deadline, "a rank did not finish"
    finally:
        for p in ctx.processes:                            # exactly the processes started here
            if p.is_alive():
                p.kill()
    res = torch.load(out)
    z = load_golden(case)
    what = f"{case} on 4 ranks (HIP kernels, gloo transport)"
    if case.startswith("fixednet"):
        torch.testing.assert_close(res["pred"], z["pred"], rtol=1e-4, atol=5e-5)
    else:
        torch.testing.assert_close(res["ent"], z["ent"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(res["loss"], z["loss"], rtol=1e-4, atol=1e-6)
    for k, g in res["grads"].items():
        assert_param_grad(z, k, g, 2e-3 if not case.startswith("fixednet") else 5e-4, 5e-6, what)
