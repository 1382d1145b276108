"""CPU oracle for the 2D->3D pose-lifting hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement of the reference algorithm
(/root/reference/phase1_lifting/baselineModel.py:14-102 and
/root/reference/phase1_lifting/train_1.py:19-23,73-100).  It exists so that
the HIP path can be checked against an independent implementation.

Rules (enforced by tests/test_layout_rules.py):
  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
    import anything from here;
  * the product package (3d_poseestimation_amd/) never imports it and has no
    CPU fallback: without the HIP extension it raises.

Parity pin: the oracle is validated against the reference model imported on
CPU in the build container (tools/make_golden.py) and against the golden
vectors that script commits under tests/golden/ (tests/test_oracle_golden.py).
"""
