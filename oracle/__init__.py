"""CPU oracle for the IHMR hot path -- TEST INFRASTRUCTURE ONLY.

Everything in this package is a plain CPU restatement (PyTorch-CPU fp32 / numpy / C) of the
reference's algorithm for the hot path named in BASELINE.json. It exists to CHECK the HIP path and
to serve as the ``cpu_baseline`` leg of ``bench.py``. Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` may import it; the product package ``ihmr_amd`` never does.

Pinning status (see DESIGN.md "Oracle"):
  * losses, Rodrigues, projection, snapshot filter/select, the OPT loop, ResNet-50 encoder, MLP
    heads, evaluator metrics: pinned against golden vectors produced by importing the reference
    itself in the build container (``tests/golden/make_golden.py``).
  * MANO LBS (third-party ``smplx==0.1.28``, absent) and the collision term (third-party
    ``sdf`` from penincillin/SDF_ihmr, unpinned upstream, absent, CUDA-only):
    **parity unpinned** -- restated from their published algorithms, anchored on the reference's
    call sites; self-consistency known-answer tests only.
"""
