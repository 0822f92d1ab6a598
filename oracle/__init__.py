"""CPU oracle for the rollout -> returns -> GRPO/PPO update hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (NumPy fp64 for
the environment dynamics, torch-CPU fp32 for the learner arithmetic) of the
reference algorithm, written from the reference's behaviour; every function
cites the reference file:line it follows (paths are relative to the reference
checkout, which never travels with this repository).

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it -- as the checker / the timed CPU baseline, never as
the thing shipped.  The product package (`trajopt-grpo_amd/`) must not import
anything from here and fails loudly when its HIP library is missing.

Parity pinning: the restatement is pinned against golden vectors generated in
the build container by importing the real reference
(`oracle/tools/gen_goldens.py` -> `tests/golden/*.npz`), plus the three
known-answer residues of the reference's own (stale) tests: the shape contract
(tests/test_rollout_manager.py:40-53), the mask-free RTG recurrence
(tests/test_rollout_buffer.py:76-92) and the upright-CartPole reward band
(tests/test_cartpole.py:91-104).
"""
