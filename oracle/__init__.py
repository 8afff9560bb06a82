"""CPU oracle for the WavJEPA pre-training step.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this
package.  The product path (`wavjepa_amd`) never does: it fails loudly when the HIP library is missing.
"""
