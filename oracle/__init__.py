"""CPU oracle for the DeepAtlas 3D volumetric hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and there only as the checker / the timed CPU baseline, never as the
thing shipped.  The product path (``deepatlas_amd``) never imports this package and
fails loudly when its HIP library is missing.

What it restates
----------------
The reference (uncbiag/DeepAtlas) is pure Python driving PyTorch (ATen) CPU kernels;
the arithmetic itself lives in PyTorch, a third-party dependency that is NOT under
/root/reference and is unpinned there (requirements.txt:1 "torch", README.md:5
"torch>=1.0").  The oracle is therefore a plain-torch, layout-explicit restatement of
the reference's modules (each function cites the reference file:line it follows),
executed on torch-CPU 2.10.0 — the same arithmetic provider the reference uses in
this image.  ``oracle/prim.c`` additionally restates the primitive ATen ops
(conv3d, transposed conv k2s2, batch-norm, max-pool, nearest up-sampling, trilinear
grid_sample, softmax) in plain C with double accumulation, from their published
definitions, as an ATen-independent cross-check.

Parity pin
----------
The reference ships no tests and no golden vectors for this path, so
the oracle is pinned against outputs of the reference itself: ``oracle/make_golden.py``
imports the reference from /root/reference in the build container, runs it on
closed-form inputs/weights and writes small fixtures to ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this restatement against every fixture.
"""
