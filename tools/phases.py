"""Phase times of a step on the main stream, streams as in a normal run (startSteps(..., profile=2))."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ccn_loader; ccn_loader.load()
from cuda_convnet_amd import CFG_DIR, layer
from cuda_convnet_amd.convnet import PRESETS, import_native
from cuda_convnet_amd.data import SyntheticDataProvider
wl = sys.argv[1] if len(sys.argv) > 1 else 'deepid2'
lib = import_native()
ldef, lpar, kind, img = PRESETS[wl]
dp = SyntheticDataProvider(kind, 128 * 16, img=img, minibatch=128, seed=1)
dims = {i: dp.get_data_dims(i) for i in range(len(dp.dims))}
layers = layer.parse_layers(os.path.join(CFG_DIR, ldef), os.path.join(CFG_DIR, lpar), dims, dp.get_num_classes(), seed=3)
lib.initModel(layers, 128, 0)
if os.environ.get('MCN_PHASES_DP'):
    lib.initDataParallel(0, 1, lib.commUniqueId(), 1)
lib.loadBatch(dp.get_next_batch()[2])
lib.startSteps(20, 0, 0); lib.finishBatch()
n = 100
lib.startSteps(n, 0, 2); lib.finishBatch()
for k, (c, ms) in sorted(lib.getProfile().items()):
    print('%-28s %8.1f us / step' % (k, ms / n * 1e3))
lib.destroyModel()
