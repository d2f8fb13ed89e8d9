"""RelGraph: the graph object handed to the operators in place of a DGLGraph.

DGL is a third-party dependency of the reference (``dgl-cuda10.0==0.5.3``,
reference README.md:16) and is not available on ROCm; the operators only need
the small protocol below, which RelGraph honours:

* callers:   ``g.nodes()``, ``g.edges(form='all')``, ``g.edata['e_type']``
             (reference models/model_lp.py:126-129, models/model_search_lp.py:135-136)
* operators: ``g.num_edges()``, ``g.edata['norm']``
             (reference models/operations_lp.py:231,318,339)
* CompGCN:   ``g.edata['etype'|'norm'|'in_edges_mask'|'out_edges_mask']``
             (reference models/compgcn.py:58,74-75)

On top of that it caches what the HIP kernels want: int32 index arrays in HBM
and a CSR-by-destination whose long rows are cut into chunks (hub nodes), so
the segmented reducers neither use float atomics nor serialise on one wave.

Edge order is the caller's and is never changed: rows [0, E/2) are original
direction edges, [E/2, E) inverse edges (both reference builders guarantee
this: train/mr_lp_train.py:80-87, utils/utils_rgcn.py:138-152).
"""
import contextlib
import os

import numpy as np
import torch

CHUNK_EDGES = 64     # in-edges reduced by one wave-group before the list is split


KNOWN_STREAMS = {}      # device index -> set of HIP streams this package launches on (functional.Fork registers them)


def register_stream(stream):
    KNOWN_STREAMS.setdefault(stream.device.index, set()).add(stream)


# Set by the static search step (sampler.static_step / supernet.SearchNetwork.static_*): plans built while it is True never talk to
# the host and settle() is a no-op (one stream).
STATIC_SHAPES = False


def settle(device):
    """Order every stream this package launches on after the work enqueued so far on the CURRENT stream of `device`
    (no-op on CPU).  Every lazily built, cached index structure ends with this: the candidates of a MixedOp run on
    several HIP streams (supernet.MixedOp), so a plan built on one stream at its first use may be read on a sibling
    stream next (found by the WN18RR full-size test: a cold first step read half-written int32 indices).  An event
    is recorded behind the build and every other known stream waits for it on the device -- no host
    synchronisation, so a launch-bound step that rebuilds its plans (a new sampled graph per step) keeps its queue
    full.  Streams created later are ordered through Fork's wait on the main stream."""
    device = torch.device(device)
    if device.type != "cuda" or STATIC_SHAPES:              # (a static step runs on ONE stream -- and may be inside a capture, where an
        return                                               #  event wait by a stream outside the capture would invalidate it)
    cur = torch.cuda.current_stream(device)
    register_stream(cur)
    others = [st for st in KNOWN_STREAMS.get(cur.device.index, ()) if st != cur]
    dflt = torch.cuda.default_stream(device)
    if dflt != cur and dflt not in others:
        others.append(dflt)
    if others:
        ev = torch.cuda.Event()
        ev.record(cur)
        for st in others:
            st.wait_event(ev)


def _device_of(val):
    if torch.is_tensor(val):
        return val.device
    if isinstance(val, dict):
        val = list(val.values())
    if isinstance(val, (list, tuple)):
        for v in val:
            d = _device_of(v)
            if d is not None:
                return d
    for attr in ("idx", "xi", "s32"):                      # GatherPlan / ComposePlan / ScorePlan
        t = getattr(val, attr, None)
        if torch.is_tensor(t):
            return t.device
    return None


def cached_on(obj, key, deps, extra, build):
    """Value derived from the tensors `deps` (and the hashable `extra`), cached on `obj` under attribute
    `key`.  The cache entry keeps the tensors themselves, so an entry can only be hit by the SAME tensor
    objects at the same in-place version (``t._version``): a different tensor that happens to reuse a freed
    address, a reassigned ``g.edata[...]`` entry or an in-place edit all rebuild."""
    slot = getattr(obj, key, None)
    if slot is not None and slot[1] == extra and len(slot[0]) == len(deps) and \
            all(a is b and (b is None or v == b._version) for (a, v), b in zip(slot[0], deps)):
        return slot[2]
    val = build()
    dev = _device_of(val)
    if dev is None:
        dev = next((d.device for d in deps if d is not None), None)
    if dev is not None:
        settle(dev)
    setattr(obj, key, ([(d, None if d is None else d._version) for d in deps], extra, val))
    return val


class _Plan(dict):
    """A plan dict whose data-dependent sizes stay in a small device tensor until someone asks for them: the HIP
    builders only enqueue work, the one device-to-host read happens at the first use of the plan."""

    def __init__(self, items, counts, names):
        super().__init__(items)
        self._counts, self._names = counts, names
        if STATIC_SHAPES:
            # static step graphs (round 5): the plan is built inside a step that is being captured into (or must stay replayable as) ONE
            # HIP graph -- launches always run over the host-known capacities (the builders pad with -1 and the kernels skip the
            # padding), nothing travels to the host, no event is queried
            self._host = self._ev = None
            return
        # the counts also travel to pinned host memory right behind the build: once that copy has landed (checked with a
        # non-blocking event query) launches switch from the padded capacity to the exact sizes without ever waiting
        self._host = torch.empty(len(names), dtype=torch.int32, pin_memory=True)
        self._host.copy_(counts, non_blocking=True)
        self._ev = torch.cuda.Event()
        self._ev.record(torch.cuda.current_stream(counts.device))

    def _resolve(self):
        if self._counts is not None:
            if self._ev is not None and self._ev.query():
                vals = self._host.tolist()
            else:
                vals = self._counts.tolist()                            # explicit request for exact sizes: the one host wait
            for n, v in zip(self._names, vals):
                dict.__setitem__(self, n, int(v))
            self._counts = self._host = self._ev = None

    def try_resolve(self):
        """True when the exact sizes are known without waiting (resolving them on the way if their copy has landed)."""
        if self._counts is None:
            return True
        if self._ev is None:                                 # built under STATIC_SHAPES: capacities only
            return False
        if self._ev.query():
            self._resolve()
            return True
        return False

    def __getitem__(self, k):
        if self._counts is not None and k in self._names:
            self._resolve()
        return dict.__getitem__(self, k)


def _hip_ready(t):
    return t.is_cuda and os.environ.get("MRG_TORCH_PLANS") != "1"


def dst_csr_plan(dst, num_nodes, chunk=CHUNK_EDGES):
    """CSR-by-destination + chunk plan for mrg_seg_reduce_fwd (include/mrgnas.h): built by the HIP plan builder
    (mrg_chunk_plan_build) for device tensors, by the tensor formulation below on the CPU (CPU tests, and the
    cross-check of the GPU tests; MRG_TORCH_PLANS=1 forces it everywhere)."""
    if _hip_ready(dst):
        return _hip_chunk_plan(dst, num_nodes, chunk)
    return dst_csr_plan_torch(dst, num_nodes, chunk)


def _hip_chunk_plan(dst, num_nodes, chunk):
    from ._lib import call, load, ptr, stream_of
    dev, E, N = dst.device, int(dst.numel()), int(num_nodes)
    d32 = dst if dst.dtype == torch.int32 else dst.to(torch.int32)
    d32 = d32.contiguous()
    i32 = lambda n: torch.empty(max(int(n), 1), dtype=torch.int32, device=dev)
    cap_c, cap_h = N + E // chunk + 1, E // chunk + 1
    eid, rowptr, deg = i32(E), i32(N + 1), i32(N)
    cn, cs, ce, csl = i32(cap_c), i32(cap_c), i32(cap_c), i32(cap_c)
    hn, hf, hc = i32(cap_h), i32(cap_h), i32(cap_h)
    counts = torch.zeros(3, dtype=torch.int32, device=dev)
    nb = load().mrg_chunk_plan_workspace_bytes(E, N)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
    call("mrg_chunk_plan_build", (ptr(d32), E, N, int(chunk), ptr(eid), ptr(rowptr), ptr(deg), ptr(cn), ptr(cs), ptr(ce), ptr(csl),
                                  ptr(hn), ptr(hf), ptr(hc), ptr(counts), ptr(ws), nb, stream_of(dst)))
    # cap_*: host-known upper bounds; the builders pad chunk_node / hub_node with -1 and the kernels skip the padding, so a
    # launch may use the capacities and never wait for the exact counts (which stay available, lazily, under n_*)
    return _Plan({"eid": eid[:E], "rowptr": rowptr, "in_degree": deg[:N], "chunk_node": cn, "chunk_start": cs, "chunk_end": ce,
                  "chunk_slot": csl, "hub_node": hn, "hub_first": hf, "hub_count": hc, "_ws": ws,
                  "cap_chunks": cap_c, "cap_hubs": cap_h, "cap_slots": cap_c},
                 counts, ("n_chunks", "n_hubs", "n_slots"))


def dst_csr_plan_torch(dst, num_nodes, chunk=CHUNK_EDGES):
    """The tensor formulation of the chunk plan (pure torch ops; runs on the tensors' device, CPU included).
    Returns a dict of int32 tensors and python ints."""
    dev = dst.device
    dst = dst.long()
    N = int(num_nodes)
    deg = torch.bincount(dst, minlength=N)
    eid = torch.argsort(dst, stable=True)                       # ascending edge ids inside each row
    rowptr = torch.zeros(N + 1, dtype=torch.long, device=dev)
    rowptr[1:] = torch.cumsum(deg, 0)
    nch = torch.clamp((deg + chunk - 1) // chunk, min=1)
    first = torch.cumsum(nch, 0) - nch
    chunk_node = torch.repeat_interleave(torch.arange(N, device=dev), nch)
    k = torch.arange(chunk_node.numel(), device=dev) - first[chunk_node]
    start = rowptr[chunk_node] + k * chunk
    end = torch.minimum(start + chunk, rowptr[chunk_node + 1])
    multi = nch[chunk_node] > 1
    slot = torch.where(multi, torch.cumsum(multi.long(), 0) - 1, torch.full_like(k, -1))
    hub_node = torch.nonzero(nch > 1).view(-1)
    hub_count = nch[hub_node]
    hub_first = torch.cumsum(hub_count, 0) - hub_count
    i32 = lambda t: t.to(torch.int32).contiguous()
    return {
        "eid": i32(eid), "rowptr": i32(rowptr), "in_degree": i32(deg),
        "chunk_node": i32(chunk_node), "chunk_start": i32(start), "chunk_end": i32(end), "chunk_slot": i32(slot),
        "hub_node": i32(hub_node), "hub_first": i32(hub_first), "hub_count": i32(hub_count),
        "n_chunks": int(chunk_node.numel()), "n_hubs": int(hub_node.numel()), "n_slots": int(multi.sum()),
    }


SPAN_ELEMS = int(os.environ.get("MRG_SPAN", "96"))      # sorted elements reduced by one lane group in the span kernels


def auto_span(E, nseg):
    """Span length of a plan: SPAN_ELEMS, longer when the segments are few and very long (DistMult's relation gradient:
    3 M scored triples over 474 relation rows) so that a hub's partial rows stay in the hundreds, while the plan keeps
    >= 8192 spans to fill the chip."""
    if nseg <= 0 or E < 16 * SPAN_ELEMS * nseg:
        return SPAN_ELEMS
    return max(SPAN_ELEMS, int(min(E / nseg / 8, E / 8192, 8 * SPAN_ELEMS)) // 32 * 32)


SPAN_SNAP = os.environ.get("MRG_SPAN_SNAP")          # lab: fixed cut displacement bound (0 = fixed spans)


def auto_snap(E, nseg, span):
    """How far a span cut may move to reach a segment boundary (span_cut in csrc/plans.hip).  Moving cuts removes the partial runs of
    short segments (no workspace slots, no hub-pass work for them) but unbalances the spans; measured (profiles/r3_north_star_spans.txt):
    where the segments are short (C5: 5 elements on average) it pays, 2.24 -> 2.14 ms; where they are long enough that cuts move
    far (FB15k-237: 19 on average, moves up to 24) it costs, 91 -> 101 us.  So: segments of up to ~2 average lengths are kept whole,
    never more than span / 4 of displacement, and nothing when the average segment is longer than span / 8."""
    if SPAN_SNAP is not None:
        return min(int(SPAN_SNAP), span // 4)
    avg = E / max(nseg, 1)
    if avg > span / 8:
        return 0
    return int(min(span // 4, max(1, round(avg))))


def span_plan(seg, nseg, span=None, snap=None):
    """Plan for mrg_span_gcs (include/mrgnas.h): elements sorted by segment, cut into spans of
    `span` consecutive sorted elements.  Only the first / last run of a span can be a partial
    segment; those get consecutive workspace slots (numbered in span order, so the slots of one
    segment are consecutive and in list order) that the hub pass adds up.  Device tensors go through the HIP
    builder (mrg_span_plan_build: histogram -> scan -> stable sort -> marking kernels), CPU tensors through the
    tensor formulation span_plan_torch."""
    span = auto_span(int(seg.numel()), int(nseg)) if span is None else span
    snap = auto_snap(int(seg.numel()), int(nseg), span) if snap is None else min(int(snap), span // 4)
    if _hip_ready(seg):
        return _hip_span_plan(seg, nseg, span, snap)
    return span_plan_torch(seg, nseg, span, snap)


def _hip_span_plan(seg, nseg, span, snap):
    from ._lib import call, load, ptr, stream_of
    dev, E, nseg = seg.device, int(seg.numel()), int(nseg)
    s32 = (seg if seg.dtype == torch.int32 else seg.to(torch.int32)).contiguous()
    n_spans = (E + span - 1) // span
    i32 = lambda n: torch.empty(max(int(n), 1), dtype=torch.int32, device=dev)
    perm, seg_s, seg_len, span_slot, span_start = i32(E), i32(E), i32(nseg), i32(2 * n_spans), i32(n_spans + 1)
    cap = 2 * n_spans + nseg
    hs, hf, hc = i32(cap), i32(cap), i32(cap)
    counts = torch.zeros(2, dtype=torch.int32, device=dev)
    nb = load().mrg_plan_workspace_bytes(E, nseg, int(span))
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
    if E == 0:
        span_start.zero_()
    call("mrg_span_plan_build", (ptr(s32), E, nseg, int(span), int(snap), ptr(perm), ptr(seg_s), ptr(seg_len), ptr(span_slot), ptr(span_start), ptr(hs),
                                 ptr(hf), ptr(hc), ptr(counts), ptr(ws), nb, stream_of(seg)))
    return _Plan({"perm": perm[:E], "seg_sorted": seg_s[:E], "seg_len": seg_len[:nseg], "span": int(span), "n_spans": int(n_spans),
                  "span_slot": span_slot, "span_start": span_start[:n_spans + 1], "hub_seg": hs, "hub_first": hf, "hub_count": hc, "E": E, "nseg": nseg,
                  "cap_hubs": cap, "cap_slots": 2 * n_spans},
                 counts, ("n_hubs", "n_slots"))


def span_plan_torch(seg, nseg, span=None, snap=None):
    """The tensor formulation of the span plan (pure torch ops; any device)."""
    dev = seg.device
    seg = seg.long()
    E, nseg = int(seg.numel()), int(nseg)
    span = auto_span(E, nseg) if span is None else span
    snap = auto_snap(E, nseg, span) if snap is None else min(int(snap), span // 4)
    perm = torch.argsort(seg, stable=True)
    seg_s = seg[perm]
    seg_len = torch.bincount(seg, minlength=nseg)
    segptr = torch.zeros(nseg + 1, dtype=torch.long, device=dev)
    segptr[1:] = torch.cumsum(seg_len, 0)
    n_spans = (E + span - 1) // span
    # cuts between spans: nominally i * span; a cut that falls inside a segment moves to the nearer end of that segment when it is at
    # most `snap` elements away (segments up to 2 * snap long are never split) -- the rule of span_cut in csrc/plans.hip
    cuts = torch.arange(n_spans + 1, device=dev) * span
    cuts[-1:] = E
    if n_spans > 1 and snap > 0:
        p = cuts[1:-1]
        s_ = seg_s[p - 1]
        b_, e_ = segptr[s_], segptr[s_ + 1]
        fwd, bwd, lim = e_ - p, p - b_, snap
        moved = torch.where(fwd <= bwd, torch.where(fwd <= lim, e_, p), torch.where(bwd <= lim, b_, p))
        cuts[1:-1] = torch.where(e_ <= p, p, moved)
    start, end = cuts[:-1], cuts[1:]
    if n_spans:
        live = end > start                                  # the last cut may have moved to E: an empty span, no runs
        f, l = seg_s[torch.clamp(start, max=max(E - 1, 0))], seg_s[torch.clamp(end - 1, min=0)]
        f_part = live & ((segptr[f] < start) | (segptr[f + 1] > end))
        l_part = live & (l != f) & (segptr[l + 1] > end)
    else:
        f = l = torch.zeros(0, dtype=torch.long, device=dev)
        f_part = l_part = torch.zeros(0, dtype=torch.bool, device=dev)
    flags = torch.stack((f_part, l_part), dim=1).reshape(-1)
    slot_ids = torch.cumsum(flags.long(), 0) - 1
    span_slot = torch.where(flags, slot_ids, torch.full_like(slot_ids, -1))
    slot_seg = torch.stack((f, l), dim=1).reshape(-1)[flags]
    if slot_seg.numel():
        hub_seg, hub_count = torch.unique_consecutive(slot_seg, return_counts=True)
    else:
        hub_seg = hub_count = torch.zeros(0, dtype=torch.long, device=dev)
    hub_first = torch.cumsum(hub_count, 0) - hub_count
    # segments without elements are appended as hubs with zero partials: the hub pass writes
    # their zero rows, so the output needs no separate zero-fill
    empty = torch.nonzero(seg_len == 0).view(-1)
    hub_seg = torch.cat((hub_seg, empty))
    hub_count = torch.cat((hub_count, torch.zeros_like(empty)))
    hub_first = torch.cat((hub_first, torch.zeros_like(empty)))
    i32 = lambda t: t.to(torch.int32).contiguous()
    return {"perm": perm, "seg_sorted": i32(seg_s), "seg_len": i32(seg_len), "span": int(span), "n_spans": int(n_spans),
            "span_slot": i32(span_slot), "span_start": i32(cuts), "hub_seg": i32(hub_seg), "hub_first": i32(hub_first), "hub_count": i32(hub_count),
            "n_hubs": int(hub_seg.numel()), "n_slots": int(slot_seg.numel()), "E": E, "nseg": nseg}


def span_meta(plan, xi, yi=None, scal=None, w_is_index=False):
    """int32x4 {seg, xi, yi, w} per sorted element (one 16-byte load each); w = float bits of scal (1.0 when absent),
    or the element's own index when `w_is_index` (per-call external scales: mrg_span_gcs ext_scal).  xi None = the
    element's own index."""
    perm = plan["perm"]
    if _hip_ready(perm) and perm.dtype == torch.int32:
        from ._lib import call, ptr, stream_of
        E = int(perm.numel())
        c32 = lambda t: None if t is None else (t if t.dtype == torch.int32 else t.to(torch.int32)).contiguous()
        xi32, yi32 = c32(xi), c32(yi)
        sc = None if scal is None else scal.float().reshape(-1).contiguous()
        meta = torch.empty(max(E, 1), 4, dtype=torch.int32, device=perm.device)
        call("mrg_span_meta_pack", (ptr(perm), ptr(plan["seg_sorted"]), ptr(xi32), ptr(yi32), ptr(sc), int(bool(w_is_index)), ptr(meta), E,
                                    stream_of(perm)))
        return meta[:E]
    perm = perm.long()
    xi_s = xi.long()[perm] if xi is not None else perm
    yi_s = yi.long()[perm] if yi is not None else torch.zeros_like(xi_s)
    if w_is_index:
        w = perm
    else:
        sc = scal.float().reshape(-1)[perm] if scal is not None else torch.ones(perm.numel(), dtype=torch.float32, device=perm.device)
        w = sc.contiguous().view(torch.int32).long()
    return torch.stack((plan["seg_sorted"].long(), xi_s, yi_s, w), dim=1).to(torch.int32).contiguous()


class RelGraph:
    """Multi-relational edge-list graph resident in HBM (see module docstring)."""

    def __init__(self, num_nodes, src, dst, etype=None, norm=None, device=None):
        dev = torch.device(device) if device is not None else torch.as_tensor(src).device
        as_idx = lambda x: torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x).to(dev).long().contiguous()
        self._n = int(num_nodes)
        self._src = as_idx(src)
        self._dst = as_idx(dst)
        if self._src.shape != self._dst.shape or self._src.dim() != 1:
            raise ValueError("src and dst must be 1-D and of equal length")
        self.edata = {}
        self.ndata = {}
        if etype is not None:
            self.edata["e_type"] = as_idx(etype)
        if norm is not None:
            self.edata["norm"] = torch.as_tensor(norm, dtype=torch.float32).to(dev)
        self._plan = None
        self._i32 = {}

    # ---- DGL-style protocol ---------------------------------------------------
    @property
    def device(self):
        return self._src.device

    def num_edges(self):
        return int(self._src.numel())

    number_of_edges = num_edges

    def number_of_nodes(self):
        return self._n

    num_nodes = number_of_nodes

    def nodes(self):
        return torch.arange(self._n, device=self.device)

    def edges(self, form="uv"):
        if form == "all":
            return self._src, self._dst, torch.arange(self.num_edges(), device=self.device)
        return self._src, self._dst

    def in_degrees(self):
        deg = getattr(self, "_in_degree32", None)          # device-built graphs carry it (mrg_build_graph)
        if deg is not None:
            return deg.long()
        if self._dst.is_cuda:                               # (torch.bincount reads its output size back to the host: not capturable)
            return torch.zeros(self._n, dtype=torch.long, device=self.device).scatter_add_(0, self._dst.long(), torch.ones_like(self._dst, dtype=torch.long))
        return torch.bincount(self._dst, minlength=self._n)

    @property
    def srcdata(self):
        return self.ndata

    @property
    def dstdata(self):
        return self.ndata

    @contextlib.contextmanager
    def local_scope(self):
        nd, ed = dict(self.ndata), dict(self.edata)
        try:
            yield
        finally:
            self.ndata.clear(); self.ndata.update(nd)
            self.edata.clear(); self.edata.update(ed)

    def to(self, device):
        device = torch.device(device)
        if device == self.device:
            return self
        g = RelGraph(self._n, self._src.to(device), self._dst.to(device))
        g.edata = {k: v.to(device) for k, v in self.edata.items()}
        g.ndata = {k: v.to(device) for k, v in self.ndata.items()}
        return g

    # ---- what the kernels consume ------------------------------------------------
    def plan(self):
        """CSR-by-destination chunk plan (built once, cached)."""
        if self._plan is None:
            self._plan = dst_csr_plan(self._dst, self._n)
            settle(self.device)
        return self._plan

    def agg_plan(self, kind):
        """Span plan + packed metadata of the destination-segmented SUM / MEAN over the edge rows
        (mean = per-element scale 1 / in-degree); cached."""
        key = "_agg_" + kind
        if key not in self._i32:
            sp = span_plan(self._dst, self._n)
            scal = None
            if kind == "mean":
                scal = (1.0 / self.in_degrees().clamp(min=1).float())[self._dst]
            meta = span_meta(sp, None, None, scal)
            settle(self.device)
            self._i32[key] = (sp, meta)
        return self._i32[key]

    def i32(self, name):
        """int32 copy of 'src' / 'dst' / an integer edata field, cached."""
        t = {"src": self._src, "dst": self._dst}.get(name)
        if t is None:
            t = self.edata[name]                 # may be reassigned / edited by the caller: tracked by identity + version
        return cached_on(self, "_i32c_" + name, (t,), None, lambda: t.to(torch.int32).contiguous())

    def bounds(self):
        """(b0, b1): rows [0, b0) are original-direction edges, [b0, b1) inverse edges.
        The reference's convention is (E/2, E) (models/operations_lp.py:318,323,327);
        relation-block shards override it with their local split."""
        E = self.num_edges()
        return E // 2, E

    def norm_flat(self):
        """edata['norm'] as a contiguous float32 [E] (the reference stores [E] or [E,1])."""
        n = self.edata["norm"]
        n = n.reshape(-1)
        if n.dtype != torch.float32 or not n.is_contiguous():
            n = n.float().contiguous()
        return n


# ---- graph construction (the step before the hot path) ----------------------------
def _deg_norm(in_deg):
    with np.errstate(divide="ignore"):
        norm = in_deg.astype(np.float32) ** np.float32(-0.5)
    norm[np.isinf(norm)] = 0
    return norm.astype(np.float32)


_DEG_NORM_TABLES = {}


def _deg_norm_table(device, need):
    """deg_norm_table[d] = float32(d) ** float32(-0.5) (0 for d = 0) on `device`, at least `need` entries: numpy's own
    values (reference utils/utils_rgcn.py:120-127), so the device edge norms are the reference's bit for bit."""
    key = str(device)
    tab = _DEG_NORM_TABLES.get(key)
    if tab is None or tab.numel() < need:
        n = max(int(need), 1 << 16, 2 * (tab.numel() if tab is not None else 0))
        tab = torch.from_numpy(_deg_norm(np.arange(n, dtype=np.int64))).to(device)
        _DEG_NORM_TABLES[key] = tab
    return tab


def build_graph_on_device(num_nodes, num_rels, triples, sorted_edges, norm_2d, device=None):
    """mrg_build_graph: the directed-edge list, its (relation, dst, src) ordering, in-degrees and edge norms computed on
    the device from a [T, 3] int64 triple tensor (host arrays are uploaded first).  Returns a RelGraph."""
    from ._lib import call, load, ptr, stream_of
    if torch.is_tensor(triples):
        tri = triples.to(device if device is not None else triples.device).long().contiguous()
    else:
        tri = torch.from_numpy(np.ascontiguousarray(np.asarray(triples, dtype=np.int64))).to(device)
    dev, T, N = tri.device, int(tri.shape[0]), int(num_nodes)
    E = 2 * T
    i64 = lambda: torch.empty(max(E, 1), dtype=torch.int64, device=dev)
    i32 = lambda n: torch.empty(max(int(n), 1), dtype=torch.int32, device=dev)
    src, dst, et = i64(), i64(), i64()
    norm = torch.empty(max(E, 1), dtype=torch.float32, device=dev)
    deg, s32, d32, e32, mx = i32(N), i32(E), i32(E), i32(E), i32(1)
    tab = _deg_norm_table(dev, E + 1)                    # an in-degree never exceeds the number of directed edges
    nb = load().mrg_build_graph_workspace_bytes(T)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
    call("mrg_build_graph", (ptr(tri), T, N, int(num_rels), int(bool(sorted_edges)), ptr(tab), int(tab.numel()), ptr(src), ptr(dst), ptr(et),
                             ptr(norm), ptr(deg), ptr(s32), ptr(d32), ptr(e32), ptr(mx), ptr(ws), nb, stream_of(tri)))
    g = RelGraph(N, src[:E], dst[:E], et[:E], norm[:E].view(-1, 1) if norm_2d else norm[:E], device=dev)
    # the int32 copies the kernels read were produced by the same launch
    cached_on(g, "_i32c_src", (g._src,), None, lambda: s32[:E])
    cached_on(g, "_i32c_dst", (g._dst,), None, lambda: d32[:E])
    cached_on(g, "_i32c_e_type", (g.edata["e_type"],), None, lambda: e32[:E])
    g._in_degree32 = deg[:N]
    return g


def _is_device(device, triples):
    if device is not None:
        return torch.device(device).type == "cuda"
    return torch.is_tensor(triples) and triples.is_cuda


def build_train_graph(num_nodes, num_rels, triples, device=None):
    """Graph of the fixed-genotype training driver (reference train/mr_lp_train.py:77-89):
    original edges then inverse edges, un-sorted; norm[e] = d_in(dst)^-1/2 * d_in(src)^-1/2, shape [E].
    With a HIP `device` (or device triples) the whole construction runs on the GPU (mrg_build_graph)."""
    if _is_device(device, triples) and os.environ.get("MRG_TORCH_PLANS") != "1":
        return build_graph_on_device(num_nodes, num_rels, triples, False, False, device)
    t = np.asarray(triples.cpu() if torch.is_tensor(triples) else triples, dtype=np.int64)
    src = np.concatenate([t[:, 0], t[:, 2]])
    dst = np.concatenate([t[:, 2], t[:, 0]])
    etype = np.concatenate([t[:, 1], t[:, 1] + num_rels])
    nn_ = _deg_norm(np.bincount(dst, minlength=num_nodes))
    return RelGraph(num_nodes, src, dst, etype, nn_[dst] * nn_[src], device=device or "cpu")


def build_search_graph(num_nodes, num_rels, triples, device=None):
    """Graph of the search driver (reference utils/utils_rgcn.py:129-158: inverse edges
    appended, then sorted by (rel, dst, src); search/mr_lp_search.py:30-36: norm shape [E,1]).
    With a HIP `device` (or device triples) the whole construction runs on the GPU (mrg_build_graph)."""
    if _is_device(device, triples) and os.environ.get("MRG_TORCH_PLANS") != "1":
        return build_graph_on_device(num_nodes, num_rels, triples, True, True, device)
    t = np.asarray(triples.cpu() if torch.is_tensor(triples) else triples, dtype=np.int64)
    src = np.concatenate([t[:, 0], t[:, 2]])
    dst = np.concatenate([t[:, 2], t[:, 0]])
    rel = np.concatenate([t[:, 1], t[:, 1] + num_rels])
    order = np.lexsort((src, dst, rel))
    src, dst, rel = src[order], dst[order], rel[order]
    nn_ = _deg_norm(np.bincount(dst, minlength=num_nodes))
    return RelGraph(num_nodes, src, dst, rel, (nn_[dst] * nn_[src]).reshape(-1, 1), device=device or "cpu")
