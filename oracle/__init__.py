"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's multigrid inner loop (emg3d/core.py kernels in C,
oracle/core_oracle.c; emg3d/solver.py driver in numpy, oracle/mg_ref.py), pinned
against the reference itself and its golden file (tests/test_oracle.py,
tools/check_oracle_vs_reference.py, tools/make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package. The emg3d_amd product path never does and fails loudly without its HIP library.
"""
