"""CPU oracle for the DiffuLab denoising hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-torch / numpy fp32 (fp64 where the reference uses fp64)
restatement of the reference algorithm for the path named in BASELINE.json's
``north_star``: DiT forward/backward under the rectified-flow and DDPM loss
heads plus the iterative sampler steps.  Every function cites the reference
``file:line`` it restates (paths relative to ``/root/reference/src/diffulab``).

Rules:
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import anything from here, and only as the checker;
  * nothing under ``diffulab_amd/`` imports it -- the product path is the HIP
    library and fails loudly when that library is missing;
  * parity is PINNED: ``tests/golden/make_golden.py`` imports the real reference
    (with import shims, in the build container only), runs it on seeded inputs
    and commits the outputs under ``tests/golden/*.npz``;
    ``tests/test_oracle_golden.py`` checks this oracle against every one of
    those vectors (the reference itself ships no tests / golden vectors).
"""
