"""CPU oracle for the UNREAL hot path.  TEST INFRASTRUCTURE ONLY.

A restatement (numpy / PyTorch-CPU) of the reference algorithm on the path
`Trainer.process()` -> maze env -> replay -> UnrealModel -> RMSProp.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this package; the product (`unreal_amd/`) never does and fails loudly if its HIP
extension is missing.

Pinning status
  * maze / pixel-change / replay (integer + byte work): PINNED against golden vectors
    produced by importing the reference's own numpy modules (tests/golden/make_fixtures.py).
  * RMSProp: PINNED by the reference's known-answer test arithmetic
    (train/rmsprop_applier_test.py:29-51) and variable-count spec (model/model_test.py:14-58).
  * NN forward / losses / gradients (TensorFlow 1.x ops, third-party, not vendored, not
    installable here): PARITY UNPINNED by the reference -- restated from model/model.py and
    TF-1.x BasicLSTMCell semantics, cross-checked by an independent numpy forward and fp64
    finite differences only.
"""
