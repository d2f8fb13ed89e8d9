#!/usr/bin/env python3
"""Benchmark of the MR-GNAS hot path on MI355X: one mixed-op supernet search step
(forward + loss + backward + clip + SGD) per "step", FB15k-237-shaped synthetic
KG, feature_dim = 200 (BASELINE.json config 2).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `value` = million directed edges of the step graph
processed per second (E / t_step / 1e6), inputs resident in HBM before timing.
`roofline` prices the libmrgnas kernel with the largest device time in the timed
steps (HIP events on the launch stream); `kernels` lists every libmrgnas entry
point of one instrumented step; `cpu_baseline` times the CPU oracle on a bounded
sample of the same workload on this host's cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

_lib = None

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate
L2_PEAK_GBS = 34500.0      # aggregate L2 bandwidth (same guide, "L2 (per XCD)"): the ceiling of gathers whose tables are cache resident
MFMA_F32_PEAK_TFS = 157.3  # dense f32-input MFMA peak (same guide)
MFMA_BF16_PEAK_TFS = 2500.0  # dense bf16 MFMA peak (same guide; not the 2:1-sparsity figure)
# The split core spends six bf16 MFMAs per f32 multiply-add tile, so its ceiling in f32-equivalent
# flops (2*rows*K*N, what `achieved` counts) is the bf16 peak / 6.
MFMA_SPLIT_PEAK_TFS = round(MFMA_BF16_PEAK_TFS / 6, 1)
MATRIX_CORE = {"mode": 0}
MFMA_BOUND = {"mrg_linear_fwd", "mrg_linear_bwd_input", "mrg_linear_bwd_weight", "mrg_dense_filter_fwd", "mrg_linear_bwd_input2",
              "mrg_dense_filter_fwd3", "mrg_linear_bwd_input3", "mrg_linear_bwd_input3_pair", "mrg_linear_bwd_weight3", "mrg_linear_relu_segmax_fwd", "mrg_linear_relu_segsum_fwd",
              "mrg_linear_relu_segreduce_fwd"}
# Entry points whose algorithmic bytes are row gathers from tables that stay resident in L2 / Infinity Cache at the
# benchmark shapes (11.6 MB entity table, 0.4 MB relation table): pricing those bytes against HBM gave "fractions"
# above 1 in round 1.  They are priced against the L2 ceiling and carry their compulsory HBM bytes separately.
L2_BOUND = {"mrg_distmult_score", "mrg_gather_compose_fwd", "mrg_zero_stats_coef", "mrg_zero_colstats"}
VALU_F32_PEAK_TFS = 157.3  # f32 vector peak (same guide): the O(D^2)-per-edge circular correlation of mrg_fused_gcs is priced against it
HBM_ACHIEVABLE_GBS = 6290.0   # the guide's measured float4 copy rate: what an HBM-bound kernel can be read against
# Entry points grouped by the DEVICE kernel that does their work (rocprofv3 kernel names in brackets): the GEMMs are spread over
# eight entry-point names, so "the entry point with the largest time" (rounds 1-2) named the wrong kernel.  `roofline` is the
# family with the largest summed time: its summed algorithmic work / its summed time.
KERNEL_FAMILIES = {
    "row_gemm [rowgemm_x3s_k rowgemm_x3q_k rowgemm_x3s8_k rowgemm_x3_k]": ["mrg_linear_fwd", "mrg_linear_bwd_input", "mrg_linear_bwd_input3", "mrg_linear_bwd_input3_pair", "mrg_dense_filter_fwd",
                                 "mrg_dense_filter_fwd3", "mrg_linear_relu_segmax_fwd", "mrg_linear_relu_segsum_fwd"],
    "weight_gradient [wgrad_x3v_k]": ["mrg_linear_bwd_weight", "mrg_linear_bwd_weight3"],
    "mixedop_epilogue [mix_colstats_k mix_fwd_k mix_bwd_reduce_k mix_bwd_apply_k]": [
        "mrg_mix_stats_coef", "mrg_mix_colstats", "mrg_mix_finalize_fwd", "mrg_mix_fwd", "mrg_mix_bwd_reduce", "mrg_mix_finalize_bwd",
        "mrg_mix_bwd_apply"],
    "span_sums [span_gcs_k]": ["mrg_span_gcs", "mrg_fused_gcs"],
    "scalar_gates [gate_row_fwd_k gate_row_bwd_k gate_fwd_k gate_bwd_k]": ["mrg_gate_collapse3", "mrg_gate_row_fwd", "mrg_gate_row_bwd", "mrg_gate_fwd", "mrg_gate_bwd",
                                                                           "mrg_gate_param_grad3"],
    "gradient_fan_in [sum_k sum_rows_gather_k]": ["mrg_sum_buffers", "mrg_sum_rows_gather"],
    "segment_reducers [seg_chunk_k seg_bwd_k segmax_bwd_gx_k]": ["mrg_seg_reduce_fwd", "mrg_seg_reduce_bwd", "mrg_seg_reduce_bwd_ordered", "mrg_seg_reduce_bwd_bits",
                                                                  "mrg_seg_reduce_heads_fwd", "mrg_segmax_bwd_input"],
    "gathers [gather_compose_k distmult_k zero_colstats_k]": ["mrg_gather_compose_fwd", "mrg_distmult_score", "mrg_zero_stats_coef", "mrg_zero_colstats"],
    "cell_zero [zero_fwd_k zero_bwd_reduce_k zero_bwd_apply_k]": ["mrg_zero_fwd", "mrg_zero_bwd_reduce", "mrg_zero_bwd_apply"],
    "compose [compose_fwd_k]": ["mrg_compose_fwd", "mrg_compose_bwd", "mrg_dense_filter_dz", "mrg_dense_filter_dz3"],
}


def family_table(stats, step_ms):
    """Per device-kernel family: launches, summed time, summed algorithmic work / summed time against the family's roofline."""
    fams = {}
    for fam, names in KERNEL_FAMILIES.items():
        recs = [stats[n] for n in names if n in stats and stats[n]["launches"] > 0 and stats[n]["ms"] > 0]
        if not recs:
            continue
        ms = sum(r["ms"] for r in recs)
        sec = ms / 1e3
        mfma = any(n in MFMA_BOUND for n in names)
        l2 = all(n in L2_BOUND for n in names if n in stats)
        if mfma:
            peak = MFMA_F32_PEAK_TFS if MATRIX_CORE["mode"] == 1 else MFMA_SPLIT_PEAK_TFS
            ach, unit, bound = sum(r["flops"] for r in recs) / sec / 1e12, "TFLOP/s", "mfma"
        else:
            ach, unit = sum(r["bytes"] for r in recs) / sec / 1e9, "GB/s"
            peak, bound = (L2_PEAK_GBS, "l2") if l2 else (HBM_PEAK_GBS, "hbm")
        row = {"bound": bound, "launches": sum(r["launches"] for r in recs), "ms_total": round(ms, 4), "achieved": round(ach, 2), "peak": peak,
               "unit": unit, "frac": round(ach / peak, 4), "share_of_step": round(ms / step_ms, 4),
               "entry_points": [n for n in names if n in stats and stats[n]["launches"] > 0]}
        if bound == "hbm":
            row["frac_achievable"] = round(ach / HBM_ACHIEVABLE_GBS, 4)      # against the guide's 6.29 TB/s copy rate
        if bound == "mfma":
            row["bytes_GBs"] = round(sum(r["bytes"] for r in recs) / sec / 1e9, 1)   # the same launches' algorithmic HBM bytes per second
        fams[fam] = row
    return fams


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="fb15k237_supernet_full",
                    choices=["fb15k237_supernet_full", "fb15k237_supernet_30k", "fb15k237_supernet_300", "wn18rr_supernet_full",
                             "fb15k237_fixed_d64", "c5_fixed_cell", "compgcn_fb15k237"])
    ap.add_argument("--caller", default="fused", choices=["fused", "reference"],
                    help="'reference': the MixedOp / cell written as the reference's models/cell_lp.py:25-33 (one operator call, "
                         "nn.BatchNorm1d, ReLU and a scaled add per candidate; gathers materialised as models/model_search_lp.py:144-145) "
                         "on this package's operators -- what the UNCHANGED reference caller gets from the operator swap alone")
    ap.add_argument("--comp-fn", default="sub", choices=["sub", "mul", "ccorr"], help="compgcn_fb15k237: the layer's composition")
    ap.add_argument("--no-caller-leg", action="store_true", help="skip the extra timed steps with the reference's literal caller")
    ap.add_argument("--dim", type=int, default=200)
    ap.add_argument("--negative", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hip-graph", action="store_true",
                    help="capture one whole step in a HIP graph and time K replays (launch-bound small step graphs; "
                         "the per-kernel events of the roofline then come from the instrumented eager step)")
    ap.add_argument("--exact-f32", action="store_true",
                    help="run every GEMM on the exact-f32 MFMA core (mrg_gemm_set_mode(1)) instead of the split-bf16 core")
    ap.add_argument("--cpu-sample", type=int, default=30000, help="graph_batch_size of the CPU-baseline sample")
    ap.add_argument("--cpu-full-graph", type=int, default=1,
                    help="1 (default): cpu_baseline.value = ONE step of the CPU oracle on the headline (full-graph) workload itself, "
                         "timed in this run (~75 s, ~71 GB of host memory); 0: the bounded sample only")
    ap.add_argument("--no-c5", action="store_true", help="skip the north-star kernel pass at the C5 shape (10 M edges, D = 256)")
    ap.add_argument("--no-exact-f32-leg", action="store_true", help="skip the extra timed steps on the exact-f32 matrix core")
    ap.add_argument("--resample", action="store_true",
                    help="sampled workloads only: draw a NEW step graph inside every timed step (device sampler, negative "
                         "sampling, graph build, index plans: reference search/mr_lp_search.py:187-214), so `value` pays for it")
    ap.add_argument("--static-step", action="store_true",
                    help="with --resample: the new step graph of every step is padded to a host-known node capacity and its node count stays "
                         "on the device (sampler.static_step, mrg_set_dynamic_rows): no host read in the step, every shape fixed -- with "
                         "--hip-graph (which implies it) the WHOLE step, sampler included, is captured once and replayed with a new draw every time")
    ap.add_argument("--comm", default=os.environ.get("MRG_COMM", "direct"), choices=["direct", "c10d", "gloo"],
                    help="N > 1: 'direct' = RCCL bound through ctypes (mr_gnas_amd/rccl.py: stream-ordered launches, the step is captured in a "
                         "HIP graph when every rank's capture succeeds; torch.distributed/gloo only bootstraps and times); 'c10d' = "
                         "torch.distributed's nccl backend for the data path (rounds 1-3)")
    ap.add_argument("--rehearse-shard", default=None, metavar="R/W",
                    help="TIMING ONLY, one GPU: run rank R of a W-way sharded step (its relation block, its node chunk, every collective "
                         "launch of the real run on a one-rank RCCL communicator; the other ranks' contributions are zeros)")
    ap.add_argument("--no-shard-graph", action="store_true", help="N > 1 with --comm direct: do not capture the sharded step in a HIP graph")
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    if a.resample and a.hip_graph:
        a.static_step = True         # a new draw per REPLAY needs fixed shapes and device-side counts: --resample --hip-graph implies --static-step
    return a


def build_step_inputs(workload, negative, seed):
    """numpy inputs of one search step: (num_ent, num_rels, node_id, graph_triples, samples, labels)."""
    from mr_gnas_amd import synth
    ds, _, size = workload.split("_")
    N, R, T = synth.SHAPES[ds]
    tri = synth.synth_kg(N, R, T, seed)
    if size == "full":
        rng = np.random.default_rng(seed + 1)
        samples, labels = synth.negative_sampling(tri, N, negative, rng)
        return N, R, np.arange(N), tri, samples, labels
    sample = {"30k": 30000, "300": 300}[size]
    node_id, gtri, samples, labels = synth.sample_step_graph(tri, sample, 0.5, negative, seed + 1)
    return N, R, node_id, gtri, samples, labels


class Step:
    """One search step on the HIP operators (reference search/mr_lp_search.py:187-245, without
    the sampler and the architect step)."""

    def __init__(self, args, device, inputs):
        from mr_gnas_amd import graph as G, supernet as S
        N, R, node_id, gtri, samples, labels = inputs
        torch.manual_seed(args.seed)
        self.g = G.build_search_graph(len(node_id), R, gtri).to(device)
        self.E = self.g.num_edges()
        src, _, _ = self.g.edges(form="all")
        self.node_id = torch.from_numpy(node_id).view(-1, 1).long().to(device)
        self.src_in = src
        self.edge_type = self.g.edata["e_type"]
        self.samples = torch.from_numpy(samples).to(device)
        self.labels = torch.from_numpy(labels).to(device)
        # reference defaults (search/mr_lp_search.py:284-326)
        self.model = S.SearchNetwork(device, N, R, 2, 1, 2, 2, args.dim, 100, 2 * R + 1, 40.0, 0.3, 0.1).to(device)
        S.xavier_init_(self.model)
        self.model.train()
        self.params, self.arch = list(self.model.parameters()), list(self.model.arch_parameters())
        self.clip = 5.0
        # clip_grad_norm_ + SGD(momentum) (reference search/mr_lp_search.py:118-119,243-245) as one library call: mr_gnas_amd/optim.py;
        # MRG_TORCH_OPTIM=1 restores torch's pair (~30 multi_tensor_apply launches)
        self.fused_opt = os.environ.get("MRG_TORCH_OPTIM", "0") != "1"
        if self.fused_opt:
            from mr_gnas_amd.optim import ClippedSGD
            self.opt = ClippedSGD(self.params, 1e-3, momentum=0.9, weight_decay=0.0, max_norm=self.clip)
        else:
            self.opt = torch.optim.SGD(self.params, 1e-3, momentum=0.9, weight_decay=0.0)
        self.last_loss = None
        self._args = args
        self.sample_size = 0
        if getattr(args, "resample", False):
            size = args.workload.split("_")[2]
            if size == "full":
                raise SystemExit("--resample needs a sampled workload (fb15k237_supernet_30k / _300)")
            self.sample_size = {"30k": 30000, "300": 300}[size]

    def resample(self, sample_size, negative):
        """A new search-step sample on the device (mr_gnas_amd.sampler: reference utils/utils_rgcn.py:79-118)."""
        from mr_gnas_amd import sampler as SM
        if getattr(self, "_kg", None) is None:
            from mr_gnas_amd import synth
            ds = self._args.workload.split("_")[0]
            N, R, T = synth.SHAPES[ds]
            self._kg = (torch.from_numpy(synth.synth_kg(N, R, T, self._args.seed)).to(self.samples.device), N, R)
            self._gen = torch.Generator(device=self.samples.device).manual_seed(self._args.seed + 7)
        tri, N, R = self._kg
        if getattr(self._args, "static_step", False):
            # capacity-padded step graph, node count on the device, no host read anywhere (sampler.static_step): every shape is the
            # same from draw to draw, so the step -- this draw included -- can be captured once and replayed (--hip-graph)
            st = SM.static_step(tri, sample_size, 0.5, R, negative, N)
            self.g, self.E = st["g"], st["g"].num_edges()
            self.node_id, self.src_in, self.edge_type = st["node_id"], st["src"], st["rel"]
            self.samples, self.labels = st["samples"], st["labels"]
            self.model.static_rows(st["n_rows"], st["n_nodes"])
            return
        g, uniq_v, src_o, rel, _, samples, labels = SM.generate_sampled_graph_and_labels(tri, sample_size, 0.5, R, negative, N,
                                                                                        generator=self._gen)
        self.g, self.E = g, g.num_edges()
        self.node_id, self.src_in, self.edge_type = uniq_v.view(-1, 1), src_o, rel
        self.samples, self.labels = samples, labels

    def __call__(self):
        if self.sample_size:
            self.resample(self.sample_size, self._args.negative)
        ent, rel = self.model(self.g, self.node_id, self.src_in, self.edge_type)
        loss = self.model.get_loss(self.g, ent, rel, self.samples, self.labels)
        loss.backward()
        if not self.fused_opt:
            torch.nn.utils.clip_grad_norm_(self.params, self.clip)      # the list is built once: walking the module tree costs ~1 ms per step
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        for a in self.arch:
            a.grad = None
        self.last_loss = loss.detach()


class FixedStep:
    """BASELINE config 1 on the GPU: one mini-batch step of the fixed README genotype (reference README.md:26,
    train/mr_lp_train.py:215-266): full training graph (un-sorted halves), feature_dim = init_fea_dim = 64, num_base_r = 23,
    batch 256, DistMult [B, N] scorer + BCELoss on label-smoothed dense targets (device LabelIndex), Adam."""

    def __init__(self, args, device, shape="fb15k237", dim=64, init_dim=64, nbase=23):
        """shape "synthetic10m" (BASELINE config 5: 10 M edges, 1 M nodes, 512 relation ids, D = 256) is SURVEY 8(d)'s
        single-cell stress: the same README-genotype step at the largest single-GPU configuration (`--workload c5_fixed_cell`)."""
        from mr_gnas_amd import graph as G, sampler as SM, supernet as S, synth
        N, R, T = synth.SHAPES[shape]
        tri = synth.synth_kg(N, R, T, args.seed)
        torch.manual_seed(args.seed)
        self.g = G.build_train_graph(N, R, tri, device=device)
        self.E = self.g.num_edges()
        geno = [S.Genotype(alpha_cell=[('pre_sub', 1, 0), ('f_sparse_comp', 2, 1), ('f_sparse_comp', 3, 2), ('a_max', 4, 2), ('a_max', 5, 3),
                                       ('f_sparse_last', 6, 5), ('f_sparse_last', 7, 5)], concat_node=[4, 5, 6, 7], score_func='sf_DisMult')]
        self.model = S.FixedNetwork(device, geno, N, R, dim, init_dim, nbase, dropout_cell=0.3, drop_aggr=0.1).to(device)
        S.xavier_init_(self.model)
        self.model.train()
        self.opt = torch.optim.Adam(self.model.parameters(), 1e-3, capturable=bool(getattr(args, "hip_graph", False)))
        idx = SM.LabelIndex(tri, R, N, device)
        rng = np.random.default_rng(args.seed + 3)
        pick = rng.integers(0, T, 256)
        self.subj = torch.from_numpy(tri[pick, 0]).to(device)
        self.rel = torch.from_numpy(tri[pick, 1]).to(device)
        self.idx, self.samples = idx, self.subj                      # `samples` only reports the batch size in the JSON
        self.last_loss = None

    def __call__(self):
        labels = self.idx.labels(self.subj, self.rel, 0.1)             # TrainDataset.__getitem__ on the device (label smoothing 0.1)
        pred = self.model(self.g, self.subj, self.rel)
        loss = F.binary_cross_entropy(pred, labels)
        loss.backward()
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        self.last_loss = loss.detach()


class CompGCNStep:
    """A 2-layer CompGCN forward + backward on the FB15k-237-shaped graph (reference models/compgcn.py:116-185: basis-decomposed
    relation table, CompGraphConv x 2 with BatchNorm, tanh and dropout; no driver of the reference reaches it, so the step is
    the module itself: forward, a squared-norm loss on both outputs, backward, Adam)."""

    def __init__(self, args, device):
        from mr_gnas_amd import compgcn as C, graph as G, synth
        N, R, T = synth.SHAPES["fb15k237"]
        tri = synth.synth_kg(N, R, T, args.seed)
        torch.manual_seed(args.seed)
        g = G.build_train_graph(N, R, tri, device=device)               # un-sorted halves: first half original, second half inverse
        self.E = g.num_edges()
        b0, _ = g.bounds()
        in_mask = torch.arange(self.E, device=device) < b0
        g.edata.update(etype=g.edata["e_type"], norm=g.edata["norm"].view(-1), in_edges_mask=in_mask, out_edges_mask=~in_mask)
        self.g = g
        self.model = C.CompGCN(100, 2 * R, N, in_dim=args.dim, layer_size=[args.dim, args.dim], comp_fn=args.comp_fn, batchnorm=True,
                               dropout=0.1, layer_dropout=[0.3, 0.3]).to(device)
        self.model.train()
        self.opt = torch.optim.Adam(self.model.parameters(), 1e-3, capturable=bool(getattr(args, "hip_graph", False)))
        self.samples = torch.empty(0, 3)
        self.last_loss = None

    def __call__(self):
        n, r = self.model(self.g)
        loss = n.square().mean() + r.square().mean()
        loss.backward()
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        self.last_loss = loss.detach()


def kernel_table(stats):
    rows = {}
    for name, r in stats.items():
        if r["launches"] == 0 or r["ms"] <= 0:
            continue
        sec = r["ms"] / 1e3
        if name in MFMA_BOUND:
            peak = MFMA_F32_PEAK_TFS if MATRIX_CORE["mode"] == 1 else MFMA_SPLIT_PEAK_TFS
            ach, unit, bound = r["flops"] / sec / 1e12, "TFLOP/s", "mfma"
        elif name in L2_BOUND:
            ach, peak, unit, bound = r["bytes"] / sec / 1e9, L2_PEAK_GBS, "GB/s", "l2"
        else:
            ach, peak, unit, bound = r["bytes"] / sec / 1e9, HBM_PEAK_GBS, "GB/s", "hbm"
        rows[name] = {"bound": bound, "launches": r["launches"], "ms_total": round(r["ms"], 4),
                      "us_per_launch": round(r["ms"] * 1e3 / r["launches"], 2), "achieved": round(ach, 2),
                      "peak": peak, "unit": unit, "frac": round(ach / peak, 4)}
    return rows


def host_cores():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota
    (the GPU boxes show 256 CPUs but grant a 16-CPU share)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    return cores


def north_star_kernel(g, dim, reps=50, tag="fb15k237"):
    """The fused per-relation gather -> compose -> segmented-sum kernel (CompGCN aggregation, reference
    models/compgcn.py:58-87) on graph `g`: segments = (destination, direction), 'sub' compose.
    bytes_alg per SURVEY section 8d: E*(8 + 4D) + 4*(nseg+1) + 4D*(R' + nseg) -- every gathered row counted per edge;
    bytes_compulsory = E*16 + 4D*(N + R' + nseg): the packed int32x4 metadata the kernel really streams, each table row
    and each output row once -- what HBM must move even when every re-read of a node row hits in cache.  At the
    FB15k-237 shape the 11.6 MB node table is cache resident, so `frac` (algorithmic) is a cache-assisted rate and
    `frac_compulsory` the true HBM share; at the C5 shape (1 GB table) the algorithmic figure is an HBM figure."""
    from mr_gnas_amd import functional as K
    src, dst, _ = g.edges(form="all")
    E, N, dev = g.num_edges(), g.number_of_nodes(), src.device
    if E == 0:
        return None
    Rp = int(g.edata["e_type"].max().item()) + 2
    b0, _ = g.bounds()
    direction = (torch.arange(E, device=dev) >= b0).long()
    cp = K.ComposePlan(src, g.edata["e_type"], dst * 2 + direction, g.norm_flat(), N, Rp, 2 * N)
    gen = torch.Generator(device=dev).manual_seed(0)
    ent = torch.randn(N, dim, device=dev, generator=gen)
    rel = torch.randn(Rp, dim, device=dev, generator=gen)
    for _ in range(3):
        K.span_gcs("sub", ent, rel, cp.m_fwd, cp.sp_seg)
    torch.cuda.synchronize()          # the plan's exact sizes have reached the host: the timed launches do not run over the padded capacity
    _lib.meter.start(["mrg_span_gcs"])
    for _ in range(reps):
        K.span_gcs("sub", ent, rel, cp.m_fwd, cp.sp_seg)
    st = _lib.meter.stop()["mrg_span_gcs"]
    sec = st["ms"] / 1e3 / st["launches"]
    nbytes = st["bytes"] / st["launches"]
    compulsory = E * 16 + 4 * dim * (N + Rp + 2 * N)
    traffic = load_traffic("north_star:" + tag)
    if tag != "c5_synthetic10m":
        # calibration point of the PMC traffic passes (tools/traffic_from_pmc.py): one streaming launch of known size,
        # 12 * D * (E + N) bytes (the step itself no longer launches mrg_compose_fwd since the gather was folded into it)
        # It doubles as the box's ACHIEVABLE streaming rate (quote it next to the 8 TB/s peak): two reads + one write.
        a_ = torch.empty(E + N, dim, device=dev)
        b_ = a_.clone()
        K.compose("sub", a_, b_)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            K.compose("sub", a_, b_)
        e1.record()
        torch.cuda.synchronize()
        stream_gbs = 5 * 12.0 * dim * (E + N) / (e0.elapsed_time(e1) / 1e3) / 1e9
        del a_, b_
    else:
        stream_gbs = None
    del cp, ent, rel
    return {"kernel": "mrg_span_gcs (CompGCN aggregation, compose=sub)", "graph": tag, "bound": "hbm", "edges": E, "segments": 2 * N,
            "streaming_kernel_GBs": None if stream_gbs is None else round(stream_gbs, 1),
            "dim": dim, "node_table_MB": round(4 * dim * N / 1e6, 1),
            "bytes_alg": int(nbytes), "bytes_compulsory": int(compulsory), "traffic": traffic,
            "us_per_launch": round(sec * 1e6, 2), "achieved": round(nbytes / sec / 1e9, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(nbytes / sec / 1e9 / HBM_PEAK_GBS, 4),
            "achieved_compulsory": round(compulsory / sec / 1e9, 1), "frac_compulsory": round(compulsory / sec / 1e9 / HBM_PEAK_GBS, 4),
            "g_edges_per_s": round(E / sec / 1e9, 3)}


def north_star_c5(device, reps=10):
    """The same kernel at BASELINE config 5's shape (10 M directed edges, 1 M nodes, 512 relation ids, D = 256): the
    1 GB node table is far beyond L2 + Infinity Cache, so the gathered rows do come from HBM."""
    from mr_gnas_amd import graph as G, synth
    N, R, T = synth.SHAPES["synthetic10m"]
    g = G.build_search_graph(N, R, synth.synth_kg(N, R, T, 0)).to(device)
    out = north_star_kernel(g, 256, reps=reps, tag="c5_synthetic10m")
    del g
    torch.cuda.empty_cache()
    return out


def load_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/*traffic*.json), or None."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json"))):
        try:
            with open(path) as f:
                t = json.load(f)
            if kernel in t.get("per_launch_bytes", {}):
                best = int(t["per_launch_bytes"][kernel])
        except Exception:
            pass
    return best


def traffic_source(kernel):
    """Which committed counter pass `load_traffic(kernel)` read (the PMC passes need rocprofv3 around the process: they are
    collected by tools/lab/profile.sh TAG traffic, not inside this run)."""
    import glob
    src = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json"))):
        try:
            with open(path) as f:
                if kernel in json.load(f).get("per_launch_bytes", {}):
                    src = os.path.relpath(path, ROOT)
        except Exception:
            pass
    return None if src is None else f"{src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of this command, not this run)"


def cpu_full_graph_record():
    """One full-graph CPU-oracle step recorded on a GPU box's host (tools/cpu_full_graph.py -> profiles/*cpu_full_graph*.json)."""
    import glob
    rec = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*cpu_full_graph*.json"))):
        try:
            with open(path) as f:
                rec = json.load(f)
            rec["source"] = os.path.relpath(path, ROOT) + " (recorded once; the default run stays within minutes)"
        except Exception:
            pass
    return rec


def _host_memory_limit_bytes():
    """What this process may allocate on the host: MemAvailable capped by the cgroup limit."""
    avail = None
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable"):
                    avail = int(line.split()[1]) * 1024
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/memory.max") as f:
            v = f.read().strip()
        if v != "max":
            cur = 0
            try:
                with open("/sys/fs/cgroup/memory.current") as f:
                    cur = int(f.read().strip())
            except Exception:
                pass
            lim = int(v) - cur
            avail = lim if avail is None else min(avail, lim)
    except Exception:
        pass
    return avail


def cpu_baseline(args, state, alphas, inputs=None):
    """The CPU oracle (a port of the reference's algorithm, oracle/) timed on this host's cores, rank 0 at N = 1 only.

    `value`: ONE supernet step on the HEADLINE workload itself -- the full step graph -- when the workload is a
    full-graph one and the host has the memory for it (~71 GB resident at fb15k237_supernet_full: 75 s on a GPU box's 16-core
    share); `sample` says so.  `sampled_30k` keeps the bounded figure rounds 1-4 reported as `value`: the median of 3 steps on a
    `--cpu-sample`-triple sampled step graph.  `--cpu-full-graph 0` (or too little host memory) falls back to the sample as `value`
    and carries the committed full-graph record instead."""
    from mr_gnas_amd import graph as G, synth
    from oracle import nets as ON
    from oracle.graph import OGraph
    ds = args.workload.split("_")[0]
    N, R, T = synth.SHAPES[ds]
    cores = host_cores()
    torch.set_num_threads(cores)
    S = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in state.items()}
    al = [a.detach().cpu().clone().requires_grad_(True) for a in alphas]

    def make_step(node_id, gtri, samples, labels):
        g = G.build_search_graph(len(node_id), R, gtri)
        src, dst, _ = g.edges(form="all")
        og = OGraph(len(node_id), src, dst, g.edata["e_type"], g.edata["norm"])
        nid, st, lt = torch.from_numpy(node_id), torch.from_numpy(samples), torch.from_numpy(labels)

        def step():
            ent, rel = ON.supernet_forward(og, S, al, nid, src, g.edata["e_type"], 2 * R + 1, 2)
            loss = ON.distmult_bce(ent, rel, st, lt)
            loss.backward()
            for v in list(S.values()) + al:
                v.grad = None
            return float(loss)
        return og, step

    # ---- the bounded sample (what rounds 1-4 reported as `value`)
    tri = synth.synth_kg(N, R, T, args.seed)
    og, step = make_step(*synth.sample_step_graph(tri, args.cpu_sample, 0.5, args.negative, args.seed + 1))
    t0 = time.perf_counter()
    step()                                   # warm-up (also bounds the cost: fewer timed steps if it is slow)
    warm = time.perf_counter() - t0
    times = []
    for _ in range(3 if warm < 12 else (1 if warm < 60 else 0)):     # median of >= 3 steps after one warm-up
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times)) if times else warm
    sampled = {"value": round(og.E / dt / 1e6, 6), "unit": "M edges/s", "seconds_per_step": round(dt, 3),
               "steps_timed": [round(t, 3) for t in times],
               "sample": f"median of {max(len(times), 1)} supernet fwd+bwd steps after 1 warm-up, sampled step graph "
                         f"graph_batch_size={args.cpu_sample} (E={og.E}, n={og.n}), D={args.dim}"}
    del og, step
    out = {"unit": "M edges/s", "cores": cores, "kind": "port", "sampled_30k": sampled}
    # ---- ONE step on the headline workload itself
    full = None
    need = 90 * 2**30                        # measured peak RSS 71 GB (profiles/r4_cpu_full_graph.json) + margin
    mem = _host_memory_limit_bytes()
    is_full = args.workload.endswith("_full") and inputs is not None
    if is_full and args.cpu_full_graph and args.dim <= 200 and (mem is None or mem >= need):
        import resource
        _, _, node_id, gtri, samples, labels = inputs
        ogf, stepf = make_step(node_id, gtri, samples, labels)
        log(f"one full-graph step of the CPU oracle (E={ogf.E}; about 75 s on 16 cores, ~71 GB resident)")
        t0 = time.perf_counter()
        loss = stepf()
        fdt = time.perf_counter() - t0
        full = {"seconds_per_step": round(fdt, 2), "value": round(ogf.E / fdt / 1e6, 6), "edges": int(ogf.E), "nodes": int(ogf.n), "steps": 1,
                "loss": loss, "peak_rss_GiB": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20, 1),
                "timed_in": "this run"}
        del ogf, stepf
    if full is not None:
        out.update(value=full["value"], seconds_per_step=full["seconds_per_step"],
                   sample=f"ONE supernet fwd+bwd step of the oracle on the headline workload itself ({args.workload}: E={full['edges']}, "
                          f"n={full['nodes']}, D={args.dim}), no warm-up, torch {torch.__version__} CPU, {cores} threads, "
                          f"{full['seconds_per_step']:.1f} s/step, peak RSS {full['peak_rss_GiB']} GiB; timed inside this run",
                   full_graph=full)
    else:
        why = ("--cpu-full-graph 0" if not args.cpu_full_graph else
               ("a sampled workload: the sample IS its size class" if not is_full else
                f"host memory available {0 if mem is None else mem / 2**30:.0f} GiB < 90 GiB needed"))
        out.update(value=sampled["value"], seconds_per_step=sampled["seconds_per_step"],
                   sample=sampled["sample"] + f"; the full-graph step was not timed in this run ({why}); "
                          "`full_graph` is the committed record of one",
                   full_graph=cpu_full_graph_record())
    return out


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product path has no CPU fallback)")
    if args.comm == "gloo":
        # LAB (not a measurement of any interconnect): N ranks that may SHARE this box's GPU(s), collectives over gloo on device
        # tensors -- the driver's N > 1 launch line, rank / world parsing, barriers, the max over ranks and rank 0's JSON line run end
        # to end on a one-GPU box:  python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 --comm gloo
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    global _lib
    from mr_gnas_amd import _lib
    lib = _lib.load()
    MATRIX_CORE["mode"] = 1 if args.exact_f32 else int(os.environ.get("MRG_GEMM_MODE", "0"))    # lab: 2 / 3 / 4 = the opt-in split-core kernels
    if lib.mrg_gemm_set_mode(MATRIX_CORE["mode"]) != 0:
        raise SystemExit("mrg_gemm_set_mode failed")

    from mr_gnas_amd import cell_lp as CL
    CL.CALLER = args.caller
    rehearse = None
    if args.rehearse_shard:
        r_, w_ = (int(v) for v in args.rehearse_shard.split("/"))
        if not (0 <= r_ < w_) or world != 1:
            raise SystemExit("--rehearse-shard R/W needs 0 <= R < W and a single process")
        rehearse = (r_, w_)
    sharded = world > 1 or os.environ.get("MRG_FORCE_SHARDED") == "1" or rehearse is not None     # the env switch rehearses the N>1 code on one GPU
    step_inputs = None
    direct = sharded and args.comm == "direct"
    comm = None
    if args.hip_graph or direct:
        # a captured step runs on ONE stream: capturing the candidate / segment side streams of the full-size step
        # segfaults inside the HIP runtime (with and without RCCL in the capture)
        # LAB: MRG_GRAPH_STREAMS=N keeps N candidate streams inside the capture (the captured graph then has parallel branches: the
        # question for launch-bound sampled steps whose kernels each fill a fraction of the chip; MRG_FORK_MIN_ROWS=0 lets them fork)
        gs = int(os.environ.get("MRG_GRAPH_STREAMS", "1"))
        os.environ["MRG_MIXED_STREAMS"] = str(gs)
        os.environ["MRG_SEGMENT_STREAMS"] = "1"
        CL.MIXED_STREAMS = gs
        from mr_gnas_amd import functional as _KF
        _KF.switches.SEGMENT_STREAMS = 1
    if sharded:
        import torch.distributed as dist
        from mr_gnas_amd import dist as MD
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29671")
        if direct:
            # control plane on gloo (rendezvous, barriers, the max over ranks of the measured time): no c10d RCCL group exists, so no
            # watchdog thread polls HIP events while a capture is open (what aborted round 3's sharded capture); data plane = rccl.py
            from mr_gnas_amd import rccl
            if rehearse is None:
                dist.init_process_group("gloo", rank=rank, world_size=world)
                barrier = dist.barrier
                # The directly bound communicator has only ever run with ONE rank (no multi-GPU node was available to the build).
                # rccl.bring_up brings it up in stages the ranks leave TOGETHER (library load -> unique id -> ncclCommInitRank -> one
                # probe all-reduce, each followed by an agreement over gloo; the two RCCL calls are bounded by a timeout): it returns a
                # communicator on every rank or None on every rank; None = every rank falls back to torch.distributed's nccl backend
                # for the data path (eager, no capture).  A rank STUCK inside RCCL ends the job with status 3 (advisor r4).
                comm = rccl.bring_up(rank, world, device, timeout_s=float(os.environ.get("MRG_RCCL_TIMEOUT", "120")), log=log)
                if comm is None:
                    log("direct RCCL bring-up failed: falling back to --comm c10d on every rank")
                    comm = dist.new_group(backend="nccl")          # the data-path group; the gloo default group keeps the control plane
                    direct = False
            else:
                comm = rccl.VirtualWorld(rehearse[0], rehearse[1], device)
                barrier = lambda: None
        elif args.comm == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
            barrier = dist.barrier
            comm = None                                            # the default (gloo) group carries the data path too
        else:
            if args.hip_graph:
                # Capturing a step that contains c10d collectives worked in round 2 and aborted in round 3 on the same settings (the
                # watchdog polls an event recorded inside the capture: timing dependent) -- opt-in, see --comm direct
                os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
                os.environ.setdefault("TORCH_NCCL_ENABLE_MONITORING", "0")
                os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "0")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            barrier = dist.barrier
        s_rank, s_world = rehearse if rehearse is not None else (rank, world)
        if args.workload == "c5_fixed_cell":                 # the fixed-genotype step on relation blocks, node tables row-sharded
            args.dim = 256
            step = MD.ShardedFixedStep(args, device, "synthetic10m", s_rank, s_world, group=comm, dim=256, init_dim=64, nbase=64)
        elif args.workload == "fb15k237_fixed_d64":
            args.dim = 64
            step = MD.ShardedFixedStep(args, device, "fb15k237", s_rank, s_world, group=comm, dim=64, init_dim=64, nbase=23)
        elif args.workload == "compgcn_fb15k237":
            raise SystemExit("compgcn_fb15k237 is a single-GPU workload")
        else:
            step = MD.ShardedStep(args, device, build_step_inputs(args.workload, args.negative, args.seed), s_rank, s_world, group=comm)
    elif args.workload == "fb15k237_fixed_d64":
        args.dim = 64
        step = FixedStep(args, device)
        barrier = lambda: None
    elif args.workload == "c5_fixed_cell":
        args.dim = 256
        step = FixedStep(args, device, shape="synthetic10m", dim=256, init_dim=64, nbase=64)
        barrier = lambda: None
    elif args.workload == "compgcn_fb15k237":
        step = CompGCNStep(args, device)
        barrier = lambda: None
    else:
        step_inputs = build_step_inputs(args.workload, args.negative, args.seed)
        step = Step(args, device, step_inputs)
        barrier = lambda: None

    log(f"inputs resident: E={getattr(step, 'E_global', step.E)} world={world}")
    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log(f"warmup step {i} done, loss {float(step.last_loss):.5f}, "
            f"HBM in use {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")

    # one instrumented step: every libmrgnas entry point bracketed by events -> decomposition table.
    # It runs on ONE stream: in the timed steps the candidates of a MixedOp share the GPU from four streams, and
    # an event pair around a kernel then also spans the time it waits for CUs held by the other streams.
    from mr_gnas_amd import functional as KF
    fork_rows = KF.switches.FORK_MIN_ROWS
    KF.switches.FORK_MIN_ROWS = 1 << 62
    _lib.meter.start()
    torch.cuda.synchronize()
    t_inst = time.perf_counter()
    step()
    raw_stats = _lib.meter.stop()                 # synchronises
    inst_ms = (time.perf_counter() - t_inst) * 1e3
    table = kernel_table(raw_stats)
    KF.switches.FORK_MIN_ROWS = fork_rows
    families = family_table(raw_stats, inst_ms)
    dom_family = max(families, key=lambda k: families[k]["ms_total"]) if families else None
    dominant = max(table, key=lambda k: table[k]["ms_total"]) if table else None

    run_step = step
    if args.resample and (sharded or (args.hip_graph and not args.static_step)):
        raise SystemExit("--resample is a single-GPU mode; replaying it from a HIP graph needs --static-step (fixed shapes, counts on the device)")
    if args.static_step and not args.resample:
        raise SystemExit("--static-step goes with --resample")
    if args.hip_graph and sharded and not direct and os.environ.get("MRG_GRAPH_SHARDED") != "1":
        # c10d collectives inside a capture: worked in round 2, aborted in round 3 (the watchdog polls an event recorded inside the
        # capture).  MRG_GRAPH_SHARDED=1 retries it; the supported way is --comm direct (the default)
        raise SystemExit("--hip-graph with --comm c10d is opt-in (MRG_GRAPH_SHARDED=1); use --comm direct")
    capture = args.hip_graph or (direct and not args.no_shard_graph)
    launch_mode = "eager"
    if capture:
        graph, ok = None, 1
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            # thread_local: another thread of the process (a c10d watchdog, when there is one) may touch the HIP runtime meanwhile
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                step()
        except Exception as e:                       # a failed capture is not fatal: the step stays eager in this same process
            if args.hip_graph and not sharded:
                raise
            ok, graph = 0, None
            log(f"capture failed ({type(e).__name__}: {str(e)[:160]}): the step stays eager")
            torch.cuda.synchronize()
        if sharded and world > 1:
            # replay only if EVERY rank captured: a rank that replays while another launches eagerly would still rendezvous inside
            # RCCL, but the agreement keeps the ranks' launch modes -- and their timings -- the same
            import torch.distributed as dist
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if ok and graph is not None:
            run_step = graph.replay
            launch_mode = "hip graph replay"
            log("one step captured in a HIP graph")

    # ---- timed region: exactly K steps, barrier + synchronize on both sides --------------------
    rows_local = int(getattr(step, "E_global", step.E)) // world + int(step.g.number_of_nodes())
    multi_stream = KF.switches.FORK_MIN_ROWS <= rows_local
    live = bool(dominant) and launch_mode == "eager" and not multi_stream
    if live:
        _lib.meter.start([dominant])
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    dom_stats = (kernel_table(_lib.meter.stop()) if live else table) if dominant else {}
    log(f"timed {args.steps} steps in {dt:.3f} s")
    if sharded and world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=("cpu" if dist.get_backend() == "gloo" else device))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    eager_ms = None
    if sharded and launch_mode != "eager":
        # the same K steps launched eagerly, so that the record carries BOTH launch modes (advisor r4: the single-GPU default is
        # eager, the sharded default replays a captured step; the N = 1 step is device-bound -- replayed 50.2 vs eager 49.9 ms,
        # DESIGN.md section 5 -- so `value` at N = 1 does not depend on the mode, at N > 1 it does)
        barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        barrier()
        e_dt = time.perf_counter() - t1
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([e_dt], dtype=torch.float64, device=("cpu" if dist.get_backend() == "gloo" else device))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e_dt = float(t.item())
        eager_ms = e_dt / args.steps * 1e3

    E_total = step.E_global if hasattr(step, "E_global") else step.E
    ms_per_step = dt / args.steps * 1e3
    value = E_total / (dt / args.steps) / 1e6

    out = {
        "metric": "million edges/sec per supernet fwd+bwd step (FB15k-237, dim=200)",
        "value": round(value, 4), "unit": "M edges/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": args.workload, "edges": int(E_total), "nodes": int(step.g.number_of_nodes()),
                   "feature_dim": args.dim, "layers": 2, "scoring_triples": int(step.samples.shape[0]),
                   "step": "supernet fwd + DistMult BCE + bwd + clip_grad_norm + SGD(momentum)",
                   "matrix_core": ("exact f32 MFMA (v_mfma_f32_32x32x2_f32)" if args.exact_f32 else
                                   "f32 via 3-way bf16 split: 6 cross terms on v_mfma_f32_32x32x16_bf16, f32 accumulate "
                                   "(error vs float64 pinned <= 1.5x the exact-f32 core in tests)"),
                   "parallelism": ("single" if world == 1 and not sharded else
                                   (f"LAB: relation-block edge shards x{world}, ranks SHARING devices, collectives over gloo on device tensors (--comm gloo): "
                                    "the N > 1 code path end to end, not a measurement of any interconnect" if args.comm == "gloo" else
                                    f"relation-block edge shards x{world} + RCCL ({'bound directly, stream-ordered' if direct else 'torch.distributed nccl backend'})")),
                   "launch": launch_mode,
                   "caller": ("cell_lp.MixedOp on the fused HIP epilogue (this package's cell_lp.py / supernet.py)" if args.caller == "fused" else
                              "the reference's literal formulation (models/cell_lp.py:25-33,95-152; models/model_search_lp.py:131-176) on this package's "
                              "operators, lazy handles " + ("on" if os.environ.get("MRG_LAZY", "1") == "1" else "off (MRG_LAZY=0)")),
                   "step_graph": (("a new sample every timed step (device sampler + negative sampling + graph build + index plans "
                                   "inside the timed region)" + ("; static shapes: node capacity padding, the draw's node count stays on the device "
                                                                  "(mrg_set_dynamic_rows), no host read" if args.static_step else ""))
                                  if args.resample else "resident, built before timing")},
        "loss": float(step.last_loss) if step.last_loss is not None else None,
    }
    # top level, next to dtype: the headline is the split-core figure; `exact_f32` below carries the all-f32-MFMA step
    out["matrix_core"] = "exact f32 MFMA" if args.exact_f32 else "3x bf16 split (6 cross terms, f32 accumulate)"
    if eager_ms is not None:
        out["eager_ms_per_step"] = round(eager_ms, 3)
        out["eager_value"] = round(E_total / eager_ms / 1e3, 4)
    if dom_family:
        # the device-kernel family that bounds the step (largest summed time in the instrumented single-stream step): summed
        # algorithmic work / summed HIP-event time of its launches; `traffic` = counter bytes of its largest entry point
        f = families[dom_family]
        big = max(f["entry_points"], key=lambda n: table[n]["ms_total"] if n in table else 0.0)
        # per-launch counter bytes exist for the launches of the default workload only (tools/lab/profile.sh profiles that command)
        counted = args.workload == "fb15k237_supernet_full" and args.dim == 200 and not sharded and args.caller == "fused"
        out["roofline"] = {"kernel": dom_family, "bound": f["bound"], "achieved": f["achieved"], "peak": f["peak"], "unit": f["unit"],
                           "frac": f["frac"], "traffic": load_traffic(big) if counted else None, "traffic_entry_point": big,
                           "traffic_source": traffic_source(big) if counted else
                           "none: the counter passes under profiles/ are of the default workload (fb15k237_supernet_full, D = 200), whose launches have other sizes",
                           "launches": f["launches"],
                           "us_per_launch": round(f["ms_total"] * 1e3 / f["launches"], 2), "ms_per_step": f["ms_total"],
                           "entry_points": f["entry_points"],
                           "timed_in": "the instrumented single-stream step before the timed steps (HIP events on the launch stream)",
                           "share_of_step": f["share_of_step"]}
        for k in ("frac_achievable", "bytes_GBs"):
            if k in f:
                out["roofline"][k] = f[k]
    if dominant and dominant in dom_stats:        # the single largest entry point, as rounds 1-2 reported it
        d = dom_stats[dominant]
        out["largest_entry_point"] = {"kernel": dominant, "bound": d["bound"], "achieved": d["achieved"], "peak": d["peak"],
                                      "unit": d["unit"], "frac": d["frac"], "launches": d["launches"], "us_per_launch": d["us_per_launch"],
                                      "timed_in": "the timed steps" if live else "the instrumented single-stream step before them"}
    out["kernel_families"] = families
    hbm_rows = [r for r in table.values() if r["bound"] == "hbm"]
    if hbm_rows:                                  # all HBM-bound entry points together
        tot_ms = sum(r["ms_total"] for r in hbm_rows)
        tot_b = sum(r["achieved"] * r["ms_total"] for r in hbm_rows)      # GB/s * ms = MB
        out["hbm_bound_total"] = {"ms_per_step": round(tot_ms, 3), "achieved": round(tot_b / tot_ms, 1), "unit": "GB/s",
                                  "frac": round(tot_b / tot_ms / HBM_PEAK_GBS, 4), "frac_achievable": round(tot_b / tot_ms / HBM_ACHIEVABLE_GBS, 4)}
    out["kernels"] = table
    fixed = args.workload in ("fb15k237_fixed_d64", "c5_fixed_cell", "compgcn_fb15k237")
    if args.workload == "compgcn_fb15k237":
        out["metric"] = f"million edges/sec per 2-layer CompGCN fwd+bwd step (FB15k-237 shape, dim={args.dim}, comp_fn={args.comp_fn})"
        out["config"]["step"] = "CompGCN(num_bases=100, layers [D, D], BatchNorm, tanh, dropout) fwd + squared-norm loss + bwd + Adam"
        out["config"]["comp_fn"] = args.comp_fn
        if "mrg_fused_gcs" in table:              # ccorr: D multiply-adds per output element and edge on the vector pipe
            r = raw_stats["mrg_fused_gcs"]
            flops = 2.0 * args.dim * args.dim * step.E * r["launches"] / max(1, r["launches"])
            sec = r["ms"] / 1e3 / r["launches"]
            out["ccorr_kernel"] = {"kernel": "mrg_fused_gcs [gcs_corr8_k]", "bound": "valu", "flop_per_launch": flops, "us_per_launch": round(sec * 1e6, 1),
                                   "achieved": round(flops / sec / 1e12, 2), "peak": VALU_F32_PEAK_TFS, "unit": "TFLOP/s",
                                   "frac": round(flops / sec / 1e12 / VALU_F32_PEAK_TFS, 4),
                                   "note": "circular correlation by its definition, 2 D^2 flop per edge (the reference's FFT form is O(D log D) but "
                                           "torch.rfft no longer exists; SURVEY 8a8)"}
    elif fixed:
        out["metric"] = ("million edges/sec per fixed-genotype train step (FB15k-237, dim=64, batch 256)" if args.workload == "fb15k237_fixed_d64"
                         else "million edges/sec per fixed-genotype train step (synthetic KG 10M edges / 1M nodes / 512 relations, dim=256, batch 256)")
        out["config"]["layers"] = 1
        out["config"]["step"] = "fixed README genotype fwd + DistMult [B,N] + BCE + bwd + Adam"
        out["config"]["hbm_peak_GiB"] = round(torch.cuda.max_memory_allocated() / 2**30, 1)
    if rehearse is not None:
        out["config"]["parallelism"] = (f"TIMING-ONLY rehearsal of rank {rehearse[0]} of {rehearse[1]} on one GPU: its relation block and node chunk, every collective "
                                        f"launch on a one-rank RCCL communicator ({getattr(comm, 'launches', 0) // max(1, args.steps + args.warmup + 2)} per step); the other ranks contribute zeros")
        out["config"]["rank_edges"] = int(step.E)
        out["value"] = None
        out["rehearsal_ms_per_step"] = round(ms_per_step, 3)
    if world == 1 and not sharded and not args.exact_f32 and not args.no_exact_f32_leg and not args.hip_graph and not fixed:
        # the same step with every GEMM on the exact-f32 MFMA pipe (v_mfma_f32_32x32x2_f32), next to the headline number
        lib.mrg_gemm_set_mode(1)
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(max(2, min(args.steps, 5))):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / max(2, min(args.steps, 5)) * 1e3
        lib.mrg_gemm_set_mode(0)
        out["exact_f32"] = {"ms_per_step": round(ms, 3), "value": round(E_total / ms / 1e3, 4), "unit": "M edges/s",
                            "matrix_core": "exact f32 MFMA (v_mfma_f32_32x32x2_f32) for every GEMM"}
    if (world == 1 and not sharded and not fixed and args.caller == "fused" and not args.no_caller_leg and not args.hip_graph
            and not args.resample):
        # the same step through the reference's literal caller formulation (supernet.SearchNetwork._forward_reference / calc_score: the
        # reference's own lines of models/model_search_lp.py:131-176 and models/cell_lp.py:25-33,95-152 on this package's operators):
        # `caller_reference` with the lazy handles of mr_gnas_amd/lazy.py (the default: what the UNCHANGED reference caller gets from the
        # one import swap), `caller_reference_eager` with MRG_LAZY=0 (eager operators: round 4's 'operator swap alone')
        from mr_gnas_amd import lazy as LZ
        CL.CALLER = "reference"
        lazy_was = LZ.ENABLED
        for key, handles in (("caller_reference", True), ("caller_reference_eager", False)):
            LZ.ENABLED = handles
            try:
                step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / 3 * 1e3
                out[key] = {"ms_per_step": round(ms, 3), "value": round(E_total / ms / 1e3, 4), "unit": "M edges/s",
                            "over_headline": round(ms / ms_per_step, 3),
                            "what": ("the reference's literal caller (per-candidate operator call + nn.BatchNorm1d + ReLU + scaled add, Python sums; gathers "
                                     "and DistMult as plain tensor indexing: models/cell_lp.py:25-33,95-152; models/model_search_lp.py:131-176) on this "
                                     "package's operators, " + ("lazy handles ON (mr_gnas_amd/lazy.py: the chain of every MixedOp evaluated as the fused "
                                                                 "epilogue, table[idx] as Gather handles)" if handles else
                                                                 "lazy handles OFF (MRG_LAZY=0: eager operators, torch BatchNorm / indexing)")),
                            "loss": float(step.last_loss)}
            except torch.cuda.OutOfMemoryError as e:
                out[key] = {"error": "out of memory: " + str(e)[:120]}
            torch.cuda.empty_cache()
        LZ.ENABLED = lazy_was
        CL.CALLER = "fused"
    if world == 1 and not sharded and args.workload not in ("c5_fixed_cell", "compgcn_fb15k237"):
        log("timing the fused compose+scatter kernel (north-star kernel) on the benchmark graph")
        out["north_star_kernel"] = north_star_kernel(step.g, args.dim, tag=args.workload.split("_")[0])
        if not args.no_c5:
            free_b = torch.cuda.mem_get_info()[0]
            if free_b > 60 * 2**30:
                log("... and at the C5 shape (10 M edges, 1 M nodes, D = 256)")
                out["north_star_kernel_c5"] = north_star_c5(device)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not fixed:
        state = step.model.state_dict()
        log("timing the CPU oracle (bounded sample, then one full-graph step)")
        out["cpu_baseline"] = cpu_baseline(args, state, step.model.arch_parameters(), step_inputs)
    if rank == 0:
        print(json.dumps(out))
    if comm is not None and getattr(comm, "is_direct_rccl", False):
        comm.destroy()
    if sharded and world > 1 or (sharded and not direct):
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
