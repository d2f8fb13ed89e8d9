"""f2: graph construction, edge ordering and the index plans, built on the device by the
HIP builders (csrc/plans.hip), must be BIT-EXACT with (1) the reference's own outputs (tests/golden/graph_small.npz,
produced by running utils/utils_rgcn.py:build_graph_from_triplets and train/mr_lp_train.py:build_graph), (2) the host
numpy builders, (3) the tensor formulations of the plans the product used before (graph.span_plan_torch /
dst_csr_plan_torch), which the CPU tests pin separately (tests/test_host_cpu.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
import mr_gnas_amd
from mr_gnas_amd import functional as K, graph as G, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _segments(E, nseg, hub, empties, seed):
    rng = np.random.default_rng(seed)
    live = max(nseg - empties, 1)
    seg = rng.integers(0, live, size=E)
    if hub and E:
        seg[: E // 3] = min(3, live - 1)
    return torch.from_numpy(seg).to(DEV)


@pytest.mark.parametrize("E,nseg,span,hub,empties", [(500, 19, 4, True, 4), (5000, 300, 96, True, 0), (100000, 29082, 96, True, 2000),
                                                       (96, 1, 96, False, 0), (97, 5, 96, False, 1), (1, 3, 96, False, 2), (0, 4, 96, False, 4),
                                                       (7, 7, 1, False, 0), (300000, 23, 96, True, 0)])
def test_span_plan_hip_equals_tensor_formulation(E, nseg, span, hub, empties):
    seg = _segments(E, nseg, hub, empties, E + nseg)
    snap = span // 4 if (E + nseg) % 2 == 0 else 0              # both forms: cuts moved to segment ends / fixed spans
    got = G.span_plan(seg, nseg, span, snap)
    ref = G.span_plan_torch(seg, nseg, span, snap)
    assert got["n_hubs"] == ref["n_hubs"] and got["n_slots"] == ref["n_slots"] and got["n_spans"] == ref["n_spans"]
    assert torch.equal(got["perm"].long(), ref["perm"].long())
    for k in ("seg_sorted", "seg_len"):
        assert torch.equal(got[k], ref[k]), k
    assert torch.equal(got["span_slot"][: 2 * ref["n_spans"]], ref["span_slot"])
    assert torch.equal(got["span_start"], ref["span_start"])
    nh = ref["n_hubs"]
    for k in ("hub_seg", "hub_first", "hub_count"):
        assert torch.equal(got[k][:nh], ref[k]), k
    # packed metadata: payload indices, float scales, index-in-w form, implicit xi
    if E:
        gen = torch.Generator(device=DEV).manual_seed(1)
        xi = torch.randint(0, 1000, (E,), device=DEV, generator=gen)
        yi = torch.randint(0, 50, (E,), device=DEV, generator=gen)
        sc = torch.randn(E, device=DEV, generator=gen)
        for args in ((xi, yi, sc, False), (xi, None, None, False), (None, None, sc, False), (xi, yi, None, True)):
            assert torch.equal(G.span_meta(got, *args), G.span_meta(ref, *args))
        # and the plan drives the kernel to the right answer
        x = torch.randn(E, 8, device=DEV, generator=gen)
        out = K.span_gcs("copy", x, None, G.span_meta(got, None), got)
        want = torch.zeros(nseg, 8, dtype=torch.float64, device=DEV).index_add_(0, seg, x.double())
        assert float((out.double() - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("E,N,chunk,hub", [(400, 23, 3, True), (400, 23, 64, True), (50000, 3000, 64, True), (0, 5, 64, False),
                                            (64, 1, 64, False), (65, 1, 64, False), (10, 40, 64, False)])
def test_chunk_plan_hip_equals_tensor_formulation(E, N, chunk, hub):
    dst = _segments(E, N, hub, min(3, N - 1), E + N + chunk)
    got = G.dst_csr_plan(dst, N, chunk)
    ref = G.dst_csr_plan_torch(dst, N, chunk)
    for k in ("n_chunks", "n_hubs", "n_slots"):
        assert got[k] == ref[k], k
    for k in ("eid", "rowptr", "in_degree"):
        assert torch.equal(got[k], ref[k]), k
    for k in ("chunk_node", "chunk_start", "chunk_end", "chunk_slot"):
        assert torch.equal(got[k][: ref["n_chunks"]], ref[k]), k
    for k in ("hub_node", "hub_first", "hub_count"):
        assert torch.equal(got[k][: ref["n_hubs"]], ref[k]), k


def _same_graph(g, ref):
    s, d, _ = g.edges(form="all")
    rs, rd, _ = ref.edges(form="all")
    assert torch.equal(s.cpu(), rs) and torch.equal(d.cpu(), rd)
    assert torch.equal(g.edata["e_type"].cpu(), ref.edata["e_type"])
    assert g.edata["norm"].shape == ref.edata["norm"].shape
    assert torch.equal(g.edata["norm"].cpu(), ref.edata["norm"])          # float32, bit for bit
    assert torch.equal(g.i32("src").cpu().long(), rs) and torch.equal(g.i32("dst").cpu().long(), rd)


def test_graph_build_on_device_matches_reference_fixture():
    z = load_golden("graph_small")
    tri = z["triples"]
    for order, fn in (("train", G.build_train_graph), ("search", G.build_search_graph)):
        for source in (tri.numpy(), tri.to(DEV)):                            # host triples are uploaded, device triples used in place
            g = fn(z["N"], z["R"], source, device=DEV)
            s, d, _ = g.edges(form="all")
            assert s.is_cuda
            assert torch.equal(s.cpu(), z[order + "/src"]) and torch.equal(d.cpu(), z[order + "/dst"])
            assert torch.equal(g.edata["e_type"].cpu(), z[order + "/etype"])
            assert torch.equal(g.edata["norm"].cpu(), z[order + "/norm"])
            assert torch.equal(g._in_degree32.cpu().long(), torch.bincount(z[order + "/dst"], minlength=z["N"]))


@pytest.mark.parametrize("ds", ["fb15k237", "wn18rr"])
def test_graph_build_on_device_full_size(ds):
    n, r, t = synth.SHAPES[ds]
    tri = synth.synth_kg(n, r, t, 0)
    _same_graph(G.build_search_graph(n, r, tri, device=DEV), G.build_search_graph(n, r, tri))
    _same_graph(G.build_train_graph(n, r, tri, device=DEV), G.build_train_graph(n, r, tri))


def test_graph_build_and_plans_at_c5_size():
    """10 M directed edges, 1 M nodes, 512 relation ids: the 49-bit (relation, dst, src) sort key, offsets beyond
    2^23, and the span plan of 2 M segments against the tensor formulation."""
    n, r, t = synth.SHAPES["synthetic10m"]
    tri = synth.synth_kg(n, r, t, 0)
    g = G.build_search_graph(n, r, tri, device=DEV)
    s, d, _ = g.edges(form="all")
    et = g.edata["e_type"]
    assert g.num_edges() == 2 * t
    key = (et * n + d) * n + s
    assert bool((key[1:] >= key[:-1]).all())                               # ordered by (relation, dst, src)
    # the same multiset of edges as the definition
    tt = torch.from_numpy(tri).to(DEV)
    want = torch.cat(((tt[:, 1] * n + tt[:, 2]) * n + tt[:, 0], ((tt[:, 1] + r) * n + tt[:, 0]) * n + tt[:, 2])).sort().values
    assert torch.equal(key, want)
    deg = torch.bincount(d, minlength=n)
    tab = torch.from_numpy(G._deg_norm(np.arange(int(deg.max()) + 1))).to(DEV)
    assert torch.equal(g.edata["norm"].view(-1), tab[deg[d]] * tab[deg[s]])
    seg = d * 2 + (torch.arange(2 * t, device=DEV) >= t).long()
    got, ref = G.span_plan(seg, 2 * n), G.span_plan_torch(seg, 2 * n)
    assert got["n_hubs"] == ref["n_hubs"] and got["n_slots"] == ref["n_slots"]
    assert torch.equal(got["perm"].long(), ref["perm"]) and torch.equal(got["span_slot"][: 2 * ref["n_spans"]], ref["span_slot"])
    assert torch.equal(got["hub_seg"][: ref["n_hubs"]], ref["hub_seg"]) and torch.equal(got["hub_count"][: ref["n_hubs"]], ref["hub_count"])
