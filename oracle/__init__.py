"""ORACLE -- test infrastructure only.

CPU float64 restatement of the reference's retargeting hot path (dex-retargeting v0.5.0,
/root/reference/src/dex_retargeting/{optimizer.py,kinematics_adaptor.py,robot_wrapper.py,seq_retarget.py}).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and only as the
checker.  The product (dex_retargeting_amd/) never imports it and has no CPU fallback.

Parity status: objective value/gradient are PINNED against the reference's own optimizer.py (imported in the
build container through oracle/ref_harness.py; vectors committed under tests/golden/).  Kinematics (pinocchio)
and the solver (nlopt SLSQP) are third-party code absent from /root/reference: PARITY UNPINNED at that
boundary -- the FK restatement is pinned by finite differences / hand-derived poses, the solver by agreement of
two independent tight minimisers (see oracle/solvers.py).
"""
