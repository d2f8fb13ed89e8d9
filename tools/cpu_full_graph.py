#!/usr/bin/env python3
"""ONE full-graph supernet step (forward + DistMult BCE + backward) of the CPU oracle on this host's cores: the CPU figure for
the very workload bench.py's headline is quoted on (fb15k237_supernet_full, E = 544 230, D = 200).  Takes minutes and ~64 GB
of RAM, so it is recorded once per round (profiles/rN_cpu_full_graph.json) and carried by bench.py as cpu_baseline.full_graph
instead of being re-run by every default bench invocation.

    python tools/cpu_full_graph.py [--out gpurun_out/cpu_full_graph.json] [--dim 200]
"""
import argparse
import json
import os
import resource
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "cpu_full_graph.json"))
    ap.add_argument("--dim", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--negative", type=int, default=10)
    a = ap.parse_args()
    from mr_gnas_amd import graph as G, supernet as S
    from oracle import nets as ON
    from oracle.graph import OGraph
    cores = bench.host_cores()
    torch.set_num_threads(cores)
    mem_kb = 0
    with open("/proc/meminfo") as f:
        for line in f:
            if line.startswith("MemTotal"):
                mem_kb = int(line.split()[1])
    limit = None
    try:
        with open("/sys/fs/cgroup/memory.max") as f:
            v = f.read().strip()
            limit = None if v == "max" else int(v)
    except Exception:
        pass
    N, R, node_id, gtri, samples, labels = bench.build_step_inputs("fb15k237_supernet_full", a.negative, a.seed)
    g = G.build_search_graph(len(node_id), R, gtri)
    src, dst, _ = g.edges(form="all")
    og = OGraph(len(node_id), src, dst, g.edata["e_type"], g.edata["norm"])
    torch.manual_seed(a.seed)
    model = S.SearchNetwork("cpu", N, R, 2, 1, 2, 2, a.dim, 100, 2 * R + 1, 40.0, 0.3, 0.1)
    S.xavier_init_(model)
    st = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    al = [p.detach().clone().requires_grad_(True) for p in model.arch_parameters()]
    nid, tri, lab = torch.from_numpy(node_id), torch.from_numpy(samples), torch.from_numpy(labels)
    print(f"[cpu_full_graph] E={og.E} n={og.n} D={a.dim} cores={cores} MemTotal={mem_kb / 2**20:.0f} GiB cgroup limit={limit}", flush=True)
    t0 = time.perf_counter()
    ent, rel = ON.supernet_forward(og, st, al, nid, src, g.edata["e_type"], 2 * R + 1, 2)
    loss = ON.distmult_bce(ent, rel, tri, lab)
    t1 = time.perf_counter()
    print(f"[cpu_full_graph] forward {t1 - t0:.1f} s, loss {float(loss):.5f}", flush=True)
    loss.backward()
    t2 = time.perf_counter()
    rec = {"workload": "fb15k237_supernet_full", "edges": int(og.E), "nodes": int(og.n), "feature_dim": a.dim, "kind": "port",
           "cores": cores, "torch": torch.__version__, "host_mem_GiB": round(mem_kb / 2**20, 1), "cgroup_mem_limit_GiB": None if limit is None else round(limit / 2**30, 1),
           "seconds_forward": round(t1 - t0, 2), "seconds_backward": round(t2 - t1, 2), "seconds_per_step": round(t2 - t0, 2),
           "value": round(og.E / (t2 - t0) / 1e6, 6), "unit": "M edges/s", "peak_rss_GiB": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20, 1),
           "steps": 1, "loss": float(loss)}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
