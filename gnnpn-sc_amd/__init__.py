"""gnnpn-sc_amd — MI355X-native (gfx950) implementation of the ML+2PN inference hot path of
wangxiaohit/GNNPN-SC behind the reference's own Python entry points.

Layout (only what the path needs):
  csrc/        hand-written HIP kernels + the C-ABI (``include/gnnpn_hip.h``) -> libgnnpn_hip.so
  _lib.py      ctypes binding of the C-ABI (fails loudly when the library is missing)
  modelML.py   ``Net``                         (mirrors reference src/models/modelML.py)
  modelPN.py   ``CombinatorialRL``, ``reward`` (mirrors reference src/models/modelPN.py)
  loadData.py  ``loadData``, ``loadDataPN``    (mirrors reference src/loadData.py)
  ML2PN.py     ``calc``, ``check``             (mirrors reference src/ML2PN.py)
  pipeline.py  device-resident end-to-end ML+2PN inference (TrainML.test + PNHigh eval block)
  dist.py      one-process-per-GPU sharding + the single all-gather of selected indices
  synth.py     seeded synthetic QWS-shaped data
"""
__version__ = "0.1.0"
