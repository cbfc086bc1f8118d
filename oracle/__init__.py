"""CPU oracle — TEST INFRASTRUCTURE ONLY.

Plain torch-fp32 (CPU) restatement of the arithmetic on the DualDiff denoising hot path.
Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
this package; nothing under `dualdiff_amd/` does.  See oracle/README.md for what is pinned
against the reference and what is not.
"""
