"""ORACLE -- test infrastructure, not product code.

CPU restatements used ONLY by tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg to
check (never to produce) results of the HIP path in ``torchsde_amd``:

  counter_brownian.c / counter.py   C twin of the counter-RNG Brownian generator (Philox + bridge tree)
  solvers_ref.py                    torch-CPU restatement of the reference's solver steps and stepping loops
  adjoint_ref.py                    torch-CPU restatement of the reference's stochastic adjoint (flat aug state)
  brownian_ref.py                   restatement of the reference's BrownianInterval (tree + LRU + seeds)

Each function cites the reference file:line it follows. Pinning (tests/golden/, generated from the real
reference by tests/golden/make_golden.py, see tests/test_oracle_*.py):
  solver / adjoint restatements  -> pinned against reference outputs and gradients under replayed increments;
  brownian_ref                   -> pinned bit-for-bit against the reference's BrownianInterval on CPU;
  bridge split / merge formulas  -> pinned against values recorded inside the reference;
  Philox                         -> pinned by Random123 known-answer vectors.
The sample path of the NEW generator for a given entropy is not (and cannot be) the reference's: the
reference's own path depends on numpy SeedSequence + torch's CPU mt19937 stream + the query history and is
pinned by no golden vector ("bitwise RNG parity: unpinned").
"""
