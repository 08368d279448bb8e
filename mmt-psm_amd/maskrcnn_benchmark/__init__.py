"""MI355X-native mirror of the reference's `maskrcnn_benchmark` operator / model API for the MMT-PSM hot path.

Same names and call signatures as Amandaynzhou/MMT-PSM's package (layers, structures, modeling,
engine.MTtrainer, `_C`), backed exclusively by hand-written gfx950 kernels in libmmtpsm.so
(mmt-psm_amd/csrc, C ABI in include/mmtpsm.h).  See DESIGN.md and INTEGRATION.md.
"""
