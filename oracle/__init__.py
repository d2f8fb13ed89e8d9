"""CPU oracle for the MR-GNAS relational message-passing hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the *checker* (or as the timed CPU
baseline) -- never as the thing shipped or measured as the GPU result.  The
product package (``mr-gnas_amd/``) does not import this package and raises if
its HIP library is missing; it has no CPU fallback.

What it is: a plain-PyTorch, float32, single-device restatement of the
reference's algorithm for the hot path, written functionally (tensors in,
tensors out, parameters passed as a ``{name: tensor}`` dict that uses the
reference's own ``state_dict`` key names).  Gradients come from autograd.
Every function cites the reference ``file:line`` it follows
(paths relative to the reference repository root).

Pinning status
--------------
* PINNED by running the reference itself in the build container
  (``tests/golden/make_golden.py`` imports ``/root/reference`` and stores
  inputs/outputs/gradients in ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
  checks this oracle against every one of those vectors):
  ``pre_*``, ``f_sparse_comp``, ``f_sparse_last`` and the other filter ops,
  the Linear+ReLU and residual parts of ``a_max/a_mean/a_sum``, the score
  functions, the fixed-genotype network and cell (``models/model_lp.py``),
  the mixed-op supernet (``models/cell_lp.py``, ``models/model_search_lp.py``),
  both graph builders' edge order / edge types / norms, and ``CompGraphConv`` /
  ``CompGCN`` for ``sub`` and ``mul``.
* PARITY UNPINNED (no reference test or runnable reference code exists):
  - DGL's reducers ``update_all(copy_e, max|sum|mean)`` and
    ``apply_edges(u_sub_e|u_mul_e)``: DGL 0.5.3 (pinned in the reference's
    README.md:16) is a third-party dependency that is not vendored and not
    installable here.  Restated from its documented semantics: rows without
    in-edges are 0; mean = sum / in-degree; the gradient of max goes to one
    arg-max edge ("lowest edge id wins" is the convention fixed here and in the
    fixture generator's stand-in).
  - ``ccorr``: the reference calls ``torch.rfft/irfft`` (removed from torch
    before the reference's own pinned version), so it cannot run; pinned to the
    direct definition ``out[k] = sum_i a[i] * b[(i+k) % D]``.
"""
