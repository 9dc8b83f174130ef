"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the attention-MIL + fusion hot path.

This package is a CPU restatement (numpy fp64 + a torch-CPU port) of the
reference's algorithm.  It exists to CHECK the HIP path; it is never the thing
shipped or measured.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.  Nothing under
`multimodalfusion_amd/` imports `oracle`.

Pinning: the reference holds no tests or golden vectors of its own
(SURVEY.md section 4), so the oracle is pinned against fixtures generated in
the build container by importing the reference's own Python modules on CPU
(`oracle/gen_golden.py` -> `tests/golden/*.npz`, script committed).
"""
