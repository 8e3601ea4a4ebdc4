"""MI355X-native compute core for the cuda-convnet / DeepID family.

Layout:
  csrc/        hand-written HIP kernels (gfx950) + the C ABI (include/mi_convnet.h)
               + the C++ layer graph and the `_ConvNet` CPython extension
  layer.py     .cfg net-definition parser (same grammar / dict schema as the reference)
  convnet.py   driver: builds the model through `_ConvNet`, train / test / features
  data.py      synthetic data providers
  cfg/         net definitions of the BASELINE configs

The directory name carries a hyphen, so import it through `ccn_loader.load()`
(repo root), which registers it as the module `cuda_convnet_amd`.
"""
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CFG_DIR = os.path.join(PKG_DIR, 'cfg')
