"""CPU-side checks of the product: the C-ABI library loads and exports every
symbol include/mrgnas.h declares, host logic (graph plan, builders,
registries) is right, and nothing silently falls back to the CPU."""
import ctypes
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
import mr_gnas_amd
from mr_gnas_amd import _lib, graph as G, operations_lp as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    declared = _lib.declared_symbols()
    assert len(declared) >= 18
    assert set(declared) == set(_lib.SIGNATURES), set(declared) ^ set(_lib.SIGNATURES)
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/mrgnas.h but not exported"
    lib = _lib.load()
    assert lib.mrg_abi_version() == _lib.ABI_VERSION
    assert lib.mrg_target_arch() == b"gfx950"
    assert b"pointer" in lib.mrg_error_string(-1)
    assert lib.mrg_gate_bwd_workspace_bytes(1000, 200) > 0


def test_argument_errors_without_gpu():
    lib = _lib.load()       # argument validation happens before any launch
    assert lib.mrg_compose_fwd(0, None, None, None, 4, 8, None) == -1
    assert lib.mrg_compose_fwd(7, ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), 4, 8, None) == -3
    assert lib.mrg_gate_fwd(ctypes.c_void_p(16), None, None, ctypes.c_void_p(16), ctypes.c_void_p(16), 5, 3, 9, 8, 1.0, None) == -2
    assert lib.mrg_seg_reduce_fwd(9, *([None] * 7), 0, None, None, None, 0, 0, None, None, None, None, 0, 8, None) == -3
    # the entry points added after the first path: shape / enum / NULL checks, size-0 calls succeed without a device
    P = ctypes.c_void_p
    assert lib.mrg_sum_buffers(None, 3, P(16), 8, 0, None) == -1                       # NULL pointer array
    assert lib.mrg_sum_buffers((ctypes.c_void_p * 1)(16), 0, P(16), 8, 0, None) == -2  # K out of range
    assert lib.mrg_sum_buffers((ctypes.c_void_p * 1)(16), 1, P(16), 0, 0, None) == 0   # nothing to do
    assert lib.mrg_distmult_score(None, None, None, None, None, None, 0, 8, None) == 0
    assert lib.mrg_distmult_score(None, P(16), P(16), P(16), P(16), P(16), 4, 8, None) == -1
    assert lib.mrg_distmult_score(P(16), P(16), P(16), P(16), P(16), P(16), 4, 0, None) == -2
    assert lib.mrg_gemm_set_mode(7) == -3 and lib.mrg_gemm_set_mode(3) == -3 and lib.mrg_gemm_set_mode(-1) == -3 and lib.mrg_gemm_set_mode(2) == 0 and lib.mrg_gemm_set_mode(0) == 0
    assert lib.mrg_gemm_workspace_bytes(400, 200) >= 400 * 200 * 4 and lib.mrg_gemm_workspace_bytes(0, 200) == 0
    assert lib.mrg_linear_fwd(P(16), P(16), None, P(16), None, 0, 8, 8, 0, None) == 0  # zero rows
    assert lib.mrg_linear_fwd(None, P(16), None, P(16), None, 4, 8, 8, 0, None) == -1
    assert lib.mrg_linear_fwd(P(16), P(16), None, P(16), None, 4, 8, 8, 5, None) == -3  # unknown activation
    assert lib.mrg_linear_bwd_input(P(16), P(16), P(16), None, 4, 8, 8, 8, 0, None) == -4  # workspace is mandatory
    with pytest.raises(_lib.MrgnasError):
        _lib.check(-2, "x")


def test_no_cpu_fallback():
    z = load_golden("ops_tiny_train")
    g = G.RelGraph(z["N"], z["src"], z["dst"], z["etype"], z["norm"])
    for name in ("pre_sub", "f_sparse_comp", "a_sum", "a_max"):
        op = O.MIXED_OPS[name]({"feature_dim": z["D"], "drop_aggr": 0.0})
        with pytest.raises(_lib.MrgnasError):
            op(g, z["x"], z["x_in"])


def test_registries_match_reference_lists():
    # reference models/operations_lp.py:8-37
    assert O.PRE_OPS == ['pre_mult', 'pre_sub', 'pre_add']
    assert O.FIRST_OPS == ['f_zero', 'f_identity', 'f_dense_comp', 'f_sparse_comp', 'f_comp']
    assert O.MIDDLE_OPS == ['a_max', 'a_sum', 'a_mean']
    assert O.LAST_OPS == ['f_zero', 'f_identity', 'f_dense_last', 'f_sparse_last']
    assert O.SF_OPS == ['sf_TransE', 'sf_DisMult']
    assert sorted(O.MIXED_OPS) == sorted(['pre_mult', 'pre_sub', 'pre_add', 'f_zero', 'f_identity', 'f_dense', 'f_dense_comp',
                                          'f_comp', 'f_sparse', 'f_sparse_comp', 'f_dense_last', 'f_sparse_last', 'a_max',
                                          'a_mean', 'a_sum'])
    assert sorted(O.MIXED_OPS_sf) == ['sf_ConvE', 'sf_DisMult', 'sf_TransE']
    z = load_golden("ops_tiny_train")
    for name in O.MIXED_OPS:          # parameter names/shapes == the reference's state_dict
        op = O.MIXED_OPS[name]({"feature_dim": z["D"], "drop_aggr": 0.0})
        tag = name if (name + "/out") in z else name + "@node"
        ref = {k[len(tag) + 7:]: tuple(v.shape) for k, v in z.items() if k.startswith(tag + "/param/")}
        assert {k: tuple(v.shape) for k, v in op.state_dict().items()} == ref, name


@pytest.mark.parametrize("chunk", [1, 3, 64])
def test_dst_csr_plan(chunk):
    rng = np.random.default_rng(0)
    N, E = 23, 400
    dst = torch.from_numpy(rng.integers(0, N - 3, size=E))          # last 3 nodes have no in-edge
    dst[:150] = 5                                                     # a hub
    p = G.dst_csr_plan(dst, N, chunk=chunk)
    eid = p["eid"].long()
    assert sorted(eid.tolist()) == list(range(E))
    assert torch.equal(p["in_degree"].long(), torch.bincount(dst, minlength=N))
    seen = torch.zeros(E, dtype=torch.long)
    nodes_seen = set()
    for c in range(p["n_chunks"]):
        v, a, b = int(p["chunk_node"][c]), int(p["chunk_start"][c]), int(p["chunk_end"][c])
        nodes_seen.add(v)
        assert 0 <= b - a <= chunk
        ids = eid[a:b]
        assert torch.all(dst[ids] == v)
        assert torch.all(ids[1:] > ids[:-1])                          # ascending edge ids inside a row
        seen[ids] += 1
        whole = (b - a) == int(p["in_degree"][v])
        assert (int(p["chunk_slot"][c]) == -1) == whole
    assert torch.all(seen == 1) and nodes_seen == set(range(N))
    slots = [int(s) for s in p["chunk_slot"] if int(s) >= 0]
    assert slots == list(range(p["n_slots"]))
    for j in range(p["n_hubs"]):
        v, f, cnt = int(p["hub_node"][j]), int(p["hub_first"][j]), int(p["hub_count"][j])
        mine = [int(p["chunk_slot"][c]) for c in range(p["n_chunks"]) if int(p["chunk_node"][c]) == v]
        assert mine == list(range(f, f + cnt)) and cnt > 1


def test_graph_builders_bit_exact_vs_reference():
    z = load_golden("graph_small")
    for order, fn in (("train", G.build_train_graph), ("search", G.build_search_graph)):
        g = fn(z["N"], z["R"], z["triples"].numpy())
        s, d, _ = g.edges(form="all")
        assert torch.equal(s, z[order + "/src"]) and torch.equal(d, z[order + "/dst"])
        assert torch.equal(g.edata["e_type"], z[order + "/etype"])
        assert torch.equal(g.edata["norm"], z[order + "/norm"])
    assert g.num_edges() == 2 * z["triples"].shape[0] and g.nodes().numel() == z["N"]


def test_harness_state_dict_keys_match_reference():
    """The reference's checkpoints must load into the harness unchanged (keys and shapes)."""
    from conftest import sub
    from mr_gnas_amd import supernet as S
    z = load_golden("supernet_tiny")
    net = S.SearchNetwork("cpu", z["Nall"], z["R"], z["layers"], 1, 2, 2, z["D"], z["D0"], z["nbase"], 9.0, 0.0, 0.0)
    ref = {**sub(z, "param/"), **sub(z, "buffer/")}
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == {k: tuple(v.shape) for k, v in ref.items()}
    net.load_state_dict(ref)
    assert [tuple(a.shape) for a in net.arch_parameters()] == [tuple(z[f"alpha/{i}"].shape) for i in range(5)]
    net.load_alpha([z[f"alpha/{i}"] for i in range(5)])
    assert repr(net.show_genotype(0)) == z["genotype0"]

    z = load_golden("fixednet_tiny")
    geno = [S.Genotype(alpha_cell=[('pre_sub', 1, 0), ('f_sparse_comp', 2, 1), ('f_sparse_comp', 3, 2), ('a_max', 4, 2),
                                   ('a_max', 5, 3), ('f_sparse_last', 6, 5), ('f_sparse_last', 7, 5)],
                       concat_node=[4, 5, 6, 7], score_func='sf_DisMult')]
    net = S.FixedNetwork("cpu", geno, z["N"], z["R"], z["D"], z["D0"], z["nbase"])
    ref = {**sub(z, "param/"), **sub(z, "buffer/")}
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == {k: tuple(v.shape) for k, v in ref.items()}


def test_library_is_not_older_than_its_sources():
    """Guards against testing a stale .so: rebuild with __graft_entry__.build() after editing csrc/."""
    import glob
    import os
    src = glob.glob(os.path.join(os.path.dirname(_lib.LIB_PATH), "..", "csrc", "*.h*")) + [_lib.HEADER_PATH]
    newest = max(os.path.getmtime(p) for p in src)
    assert os.path.getmtime(_lib.LIB_PATH) >= newest, "libmrgnas_hip.so is older than its sources: run __graft_entry__.build()"


@pytest.mark.parametrize("span", [1, 4, 64, 7, 33])
def test_span_plan_emulation(span):
    """Emulate mrg_span_gcs on the CPU from the plan alone and compare with index_add."""
    rng = np.random.default_rng(span)
    nseg, E, D = 19, 500, 3
    seg = torch.from_numpy(rng.integers(0, nseg - 4, size=E))          # last segments stay empty
    seg[:200] = 3                                                        # a hub spanning many spans
    x = torch.randn(E, D, dtype=torch.float64)
    snap = span // 4
    p = G.span_plan(seg, nseg, span=span, snap=snap)
    meta = G.span_meta(p, torch.arange(E), None, None)
    assert meta.shape == (E, 4) and meta.dtype == torch.int32
    assert torch.equal(meta[:, 0], p["seg_sorted"]) and torch.all(meta[1:, 0] >= meta[:-1, 0])
    assert meta[:, 3].view(torch.float32).eq(1.0).all()
    out = torch.zeros(nseg, D, dtype=torch.float64)
    ws = torch.zeros(max(p["n_slots"], 1), D, dtype=torch.float64)
    written = torch.zeros(nseg, dtype=torch.long)
    cuts = p["span_start"].tolist()
    assert cuts[0] == 0 and cuts[-1] == E and all(x < y for x, y in zip(cuts, cuts[1:-1])) and cuts[-2] <= cuts[-1]   # only the LAST span may be empty
    # a cut is nominal (i * span) or moved by at most snap (= span / 4 here) to a segment boundary; no segment of length <= 2 * snap is ever split
    segptr = np.concatenate(([0], np.cumsum(np.bincount(seg.numpy(), minlength=nseg))))
    for i, c in enumerate(cuts[1:-1], start=1):
        assert abs(c - i * span) <= snap and (c == i * span or c in segptr)
    for v in range(nseg):
        if 0 < segptr[v + 1] - segptr[v] <= 2 * snap:
            assert not any(segptr[v] < c < segptr[v + 1] for c in cuts), f"segment {v} of length <= span / 2 is split"
    for sp in range(p["n_spans"]):
        a, b = cuts[sp], cuts[sp + 1]
        if a >= b:
            continue
        sf, sl_ = int(p["span_slot"][2 * sp]), int(p["span_slot"][2 * sp + 1])
        acc, cur, first = torch.zeros(D, dtype=torch.float64), int(meta[a, 0]), True

        def flush(slot):
            if slot >= 0:
                ws[slot] = acc
            else:
                out[cur] = acc
                written[cur] += 1
        for j in range(a, b):
            if int(meta[j, 0]) != cur:
                flush(sf if first else -1)
                acc, cur, first = torch.zeros(D, dtype=torch.float64), int(meta[j, 0]), False
            acc = acc + x[int(meta[j, 1])]
        flush(sf if first else sl_)
    out += 7.0 * (written == 0).view(-1, 1)            # the kernel's output buffer is NOT pre-zeroed
    for h in range(p["n_hubs"]):
        s0, cnt, v = int(p["hub_first"][h]), int(p["hub_count"][h]), int(p["hub_seg"][h])
        assert written[v] == 0
        out[v] = ws[s0:s0 + cnt].sum(0)
        written[v] += 1
    ref = torch.zeros(nseg, D, dtype=torch.float64).index_add(0, seg, x)
    torch.testing.assert_close(out, ref, rtol=1e-12, atol=1e-12)
    assert torch.all(written == 1)                      # every segment written exactly once, empty ones by the hub pass


def test_async_fill_kernels_do_not_spill():
    """The split-core kernels fill registers asynchronously from inline asm (gemm_x3.hpp); that contract only
    holds while hipcc keeps those values in registers.  Compile the two translation units that instantiate them
    with the resource-usage remarks and require zero VGPR spills for every *_x3_k instance (a spilled build
    measured wrong results in the lab)."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "mr-gnas_amd", "csrc")
    seen = 0
    for unit in ("linear.hip", "dense.hip"):
        out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage",
                              "--cuda-device-only", "-c", os.path.join(csrc, unit), "-o", os.devnull],
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        name = None
        for line in out.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
            m = re.search(r"VGPRs Spill: (\d+)", line)
            if m and name and ("_x3_k" in name or "_x3s_k" in name or "wgrad_x3" in name):
                seen += 1
                # the LDS-weight kernel's gate epilogue (EPI_GATE = 1: three row pointers per accumulator row next to 112
                # accumulators) spills 8 registers since round 3 -- inside the epilogue, after every asynchronous fill has been
                # waited for (vmcnt(0) in front of it), and its results are bit-identical with the spill-free one-wave kernel
                # (test_split_core_kernels_are_bit_exact_with_each_other); bounded here so that it cannot grow unnoticed
                gate_x3s = re.search(r"rowgemm_x3s_kILi\d+ELi1E", name) is not None
                assert int(m.group(1)) <= (16 if gate_x3s else 0), f"{name} spills {m.group(1)} VGPRs"
    assert seen >= 20


def test_fan_sums_reader_gradients_and_tolerates_unused_aliases():
    """functional.Fan: aliases of one tensor for its readers; the gradient is the sum over the readers that were
    used, unused aliases contribute nothing, and more readers than one batch of aliases chain correctly."""
    from mr_gnas_amd import functional as K
    x = torch.randn(5, 3, requires_grad=True)
    fan = K.Fan(x, 40)
    weights = [float(i + 1) for i in range(30)]            # 30 readers > Fan.BATCH: exercises the chained batches
    total = sum(w * fan.take().sum() for w in weights)
    fan.take()                                              # an alias nobody reads
    total.backward()
    assert torch.allclose(x.grad, torch.full_like(x, sum(weights)))
    with pytest.raises(RuntimeError):
        small = K.Fan(x, 2)
        small.take(); small.take(); small.take()
    y = torch.randn(4, 2)                                   # no gradient needed: the tensor itself is handed out
    assert K.Fan(y, 8).take() is y


def test_plan_caches_follow_tensor_identity_not_addresses():
    """Advisor r1 (high / medium): index plans are cached by the identity + in-place version of the tensors they
    were derived from.  A new same-shape batch that lands on a freed batch's address, a reassigned edata entry or
    an in-place edit must all rebuild the plan."""
    from mr_gnas_amd import supernet as S

    built = []

    class Holder:
        pass

    h = Holder()

    def plan_for(t):
        return G.cached_on(h, "_c", (t,), (7,), lambda: built.append(t.clone()) or len(built))

    seen_ptrs = set()
    for i in range(4):                                       # function-scoped batches: the allocator recycles the address
        batch = torch.full((64, 3), i, dtype=torch.int64)
        seen_ptrs.add(batch.data_ptr())
        n = plan_for(batch)
        assert n == i + 1 and int(built[-1][0, 0]) == i     # never the previous batch's plan
        assert plan_for(batch) == n                          # same object, same version: hit
        del batch
    keep = torch.zeros(8, 3, dtype=torch.int64)
    n = plan_for(keep)
    keep[0, 0] = 5                                           # in-place edit bumps _version
    assert plan_for(keep) == n + 1
    assert G.cached_on(h, "_c", (keep,), (8,), lambda: "other-extra") == "other-extra"

    # RelGraph.i32 follows a reassigned edata entry
    g = G.RelGraph(4, [0, 1, 2], [1, 2, 3], etype=[0, 1, 0], norm=[1.0, 1.0, 1.0])
    assert g.i32("e_type").tolist() == [0, 1, 0]
    g.edata["e_type"] = torch.tensor([1, 1, 1])
    assert g.i32("e_type").tolist() == [1, 1, 1]

    # SearchNetwork.prepare: a graph object reused with other index tensors gets new gather plans
    net = S.SearchNetwork("cpu", 10, 2, 1, 1, 2, 2, 8, 8, 5, 9.0, 0.0, 0.0)
    nid = torch.arange(4).view(-1, 1)
    src, _, _ = g.edges(form="all")
    p1 = net.prepare(g, nid, src, g.edata["e_type"])
    assert net.prepare(g, nid, src, g.edata["e_type"]) is p1
    nid2 = torch.tensor([3, 2, 1, 0]).view(-1, 1)
    p2 = net.prepare(g, nid2, src, g.edata["e_type"])
    assert p2 is not p1 and p2[0].idx.tolist() != p1[0].idx.tolist()
    assert hasattr(net, "_loss")                             # what the reference's Architect calls


def test_sync_batch_norm_eval_uses_running_statistics():
    """Advisor r1 (low): dist.sync_batch_norm must behave like nn.BatchNorm1d in eval mode (no batch statistics,
    no collective)."""
    from mr_gnas_amd import dist as MD
    bn = torch.nn.BatchNorm1d(5)
    bn.running_mean.copy_(torch.randn(5))
    bn.running_var.copy_(torch.rand(5) + 0.5)
    bn.eval()
    x = torch.randn(7, 5)
    torch.testing.assert_close(MD.sync_batch_norm(x, bn, 7, None), bn(x))


def test_rccl_binding_loads_and_matches_the_header():
    """mr_gnas_amd/rccl.py binds the librccl.so PyTorch ships (no GPU needed to load it): every entry point the sharded step launches
    is there, the unique id is the header's 128 opaque bytes, the dtype / op codes are the header's enum values
    (/opt/rocm/include/rccl/rccl.h: ncclFloat32 = 7, ncclFloat64 = 8, ncclInt32 = 2, ncclSum = 0, ncclMax = 2)."""
    import ctypes
    from mr_gnas_amd import rccl
    lib = rccl.load()
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllReduce", "ncclReduceScatter", "ncclAllGather", "ncclGetErrorString"):
        assert hasattr(lib, name), name
    assert ctypes.sizeof(rccl._UniqueId) == 128
    assert rccl._DTYPE[torch.float32] == 7 and rccl._DTYPE[torch.float64] == 8 and rccl._DTYPE[torch.int32] == 2
    assert rccl._OP == {"sum": 0, "max": 2, "min": 3}
    assert lib.ncclGetErrorString(0).decode().lower().startswith("no error")


def test_lazy_handles_random_programs_equal_eager():
    """mr_gnas_amd/lazy.py is observational: whatever torch code touches an operator's lazy handle, the values (and the gradients)
    are those of the eager evaluation.  Random programs over the handle vocabulary (operator call, BatchNorm, ReLU, dropout, scaling,
    sums) mixed with calls OUTSIDE it (views, cat / stack, reductions, masked indexing, handle x handle products, in-place updates,
    detach, no_grad, clone, comparison) run twice -- handles on (lazy.FORCE_CPU: CPU restatements of the operators behind this
    package's forward / run protocol, every handle evaluated literally) and off -- and every observed tensor must be bit-equal."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_kernels as CK
    from mr_gnas_amd import lazy as LZ, operations_lp as OPS

    def wrap(inner, node_rows):
        class W(OPS._Operator):
            def __init__(self):
                super().__init__()
                for k, m in inner.named_children():
                    setattr(self, k, m)
                object.__setattr__(self, "_inner", inner)

            def out_shape(self, g, src_emb):
                return (g.number_of_nodes() if node_rows else src_emb.shape[0], src_emb.shape[1])

            def run(self, g, a, b, for_epilogue=False):
                return inner(g, LZ.real(a), LZ.real(b))
        return W()

    rng = np.random.default_rng(11)
    n, R, T, D = 30, 4, 120, 8
    tri = np.stack([rng.integers(0, n, T), rng.integers(0, R, T), rng.integers(0, n, T)], 1)
    g = G.build_search_graph(n, R, tri)
    E = g.num_edges()
    reg = CK.registry()
    torch.manual_seed(0)
    edge_ops = [wrap(reg[k]({"feature_dim": D}), False) for k in ("f_dense_comp", "f_sparse_comp", "f_comp", "f_identity")]
    node_ops = [wrap(reg[k]({"feature_dim": D, "drop_aggr": 0.0}), True) for k in ("a_max", "a_sum", "a_mean")]
    bns = [torch.nn.BatchNorm1d(D) for _ in range(3 + 14)]       # 0-2: reused by the MixedOp idiom (always on handles); 3 + step: once per program step
    params = [p for m in edge_ops + node_ops + bns for p in m.parameters()]
    state0 = [t.clone() for m in bns for t in m.state_dict().values()]
    # (a BatchNorm module applied twice in one forward, once to a handle and once to a plain tensor, updates its running statistics in
    #  evaluation order, not call order -- lazy.py's one documented difference; the reference uses every BatchNorm once per forward)

    def program(seed, handles):
        LZ.ENABLED, LZ.FORCE_CPU = handles, handles
        for m, i in zip(bns, range(len(bns))):
            m.load_state_dict(dict(zip(m.state_dict().keys(), state0[i * 5:(i + 1) * 5])))
            m.train()
        for p in params:
            p.grad = None
        r = np.random.default_rng(seed)
        gen = torch.Generator().manual_seed(seed)
        x = torch.randn(E + n, D, generator=gen).requires_grad_(True)
        y = torch.randn(E + n, D, generator=gen).requires_grad_(True)
        w = torch.rand(6, generator=gen).requires_grad_(True)
        edge, node, seen = [x], [], []                       # tensors / handles with E + n rows, with n rows; what the program looked at

        def obs(t):
            seen.append(torch.as_tensor(t).detach().clone() if isinstance(t, torch.Tensor) else torch.tensor(float(t)))

        for step in range(14):
            a = int(r.integers(0, 16))
            pool = edge if (not node or r.random() < 0.6) else node
            h = pool[int(r.integers(0, len(pool)))]
            if a <= 2:                                        # an operator call on the latest states
                op = edge_ops[int(r.integers(0, len(edge_ops)))]
                edge.append(op(g, edge[-1], edge[0]))
            elif a == 3:
                node.append(node_ops[int(r.integers(0, len(node_ops)))](g, edge[-1], edge[0]))
            elif a == 4:                                      # the MixedOp idiom: w * relu(bn(op(...))) summed
                ops = [edge_ops[int(i)] for i in r.integers(0, len(edge_ops), 3)]
                edge.append(sum(w[k] * torch.relu(bns[k](op(g, edge[-1], edge[0]))) for k, op in enumerate(ops)))
            elif a == 5:
                pool.append(torch.nn.functional.dropout(torch.relu(bns[3 + step](h)), 0.0, True))
            elif a == 6:
                pool.append(h + pool[0] if pool is edge else h + h)
            elif a == 7:                                      # outside the vocabulary: views and reductions
                obs(h.t().contiguous().sum(1))
                obs(h.view(-1)[::7])
            elif a == 8:
                obs(torch.cat((h, h * 2.0), 1).mean(0))
                obs(torch.stack((h, h)).amax(0))
            elif a == 9:
                m_ = h.detach() > 0.3
                obs(h[m_])
                obs((h > 0).sum())
            elif a == 10:
                pool.append(h * h)                            # handle x handle
            elif a == 11:
                with torch.no_grad():
                    obs(h.clone().add_(1.0).mul_(0.5))
                obs(h.detach().abs().max())
            elif a == 12:
                z = h.clone()
                z[0] = 0.0                                     # in-place write into a copy
                pool.append(z)
            elif a == 13:
                obs(h.shape[0] * 1.0)
                obs(h.dim())
                obs(h.new_zeros(3).sum() + h.sum())
            elif a == 14:
                pool.append(torch.where(h > 0, h, 0.1 * h))
            else:
                pool.append(0.5 * h - pool[0] if pool is edge else -h)
        loss = sum(t.square().mean() for t in edge[1:]) + sum(t.abs().mean() for t in node)
        if isinstance(loss, torch.Tensor) and loss.requires_grad:
            loss.backward()
        obs(loss)
        grads = [None if p.grad is None else p.grad.clone() for p in [x, y, w] + params]
        stats = [t.clone() for m in bns for t in m.state_dict().values()]
        return seen, grads, stats, sum(isinstance(t, LZ.Lazy) for t in edge + node)

    try:
        handles_seen = 0
        for seed in range(int(os.environ.get("MRG_LAZY_FUZZ", "60"))):
            lazy_run = program(seed, True)
            eager_run = program(seed, False)
            handles_seen += lazy_run[3]
            assert eager_run[3] == 0
            assert len(lazy_run[0]) == len(eager_run[0])
            for i, (a, b) in enumerate(zip(lazy_run[0], eager_run[0])):
                assert a.shape == b.shape and torch.equal(a, b), f"seed {seed}: observation {i} differs by {float((a - b).abs().max()):.3e}"
            for a, b in zip(lazy_run[2], eager_run[2]):
                assert torch.equal(a, b), f"seed {seed}: BatchNorm running statistics differ"
            for i, (a, b) in enumerate(zip(lazy_run[1], eager_run[1])):
                assert (a is None) == (b is None), f"seed {seed}: gradient {i} present on one side only"
                if a is not None:                              # (the order in which autograd sums a leaf's gradients may differ: rounding)
                    assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 1e-6, f"seed {seed}: gradient {i}"
        assert handles_seen > 100                               # the programs did run on handles
    finally:
        LZ.ENABLED, LZ.FORCE_CPU = True, False
