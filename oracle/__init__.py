"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT.

CPU restatement of the reference (Amandaynzhou/MMT-PSM) algorithm for the hot path named in
BASELINE.json: used only as the checker by tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg.  Nothing under mmt-psm_amd/ may import this package.

  native/mmt_oracle.c  plain C: NMS, ROIAlign fwd/bwd, polygon rasteriser
  native.py            ctypes binding of the above
  model.py             pure-torch (CPU, NCHW) restatement of GeneralizedRCNN.forward /
                       forward_teacher / forward_student, the PSM and MGD losses, loss weighting
                       and the EMA update
  refharness/          build-container-only: imports the REAL reference to pin the oracle and to
                       emit tests/golden/*.npz  (never runs on the GPU box)
  _ref/                git-ignored build output: the reference's own `_C` CPU extension

Parity status: PINNED.  The reference ships no golden vectors for this path,
so the oracle is pinned against outputs of the reference itself run in the build container
(tests/golden/gen_golden.py -> tests/golden/*.npz; tests/test_oracle_golden.py).
"""
