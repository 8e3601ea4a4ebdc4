"""Registers the package directory `cuda-convnet_amd/` as module `cuda_convnet_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))


def load():
    if 'cuda_convnet_amd' in sys.modules:
        return sys.modules['cuda_convnet_amd']
    pkg = os.path.join(ROOT, 'cuda-convnet_amd')
    spec = importlib.util.spec_from_file_location('cuda_convnet_amd', os.path.join(pkg, '__init__.py'),
                                                  submodule_search_locations=[pkg])
    mod = importlib.util.module_from_spec(spec)
    sys.modules['cuda_convnet_amd'] = mod
    spec.loader.exec_module(mod)
    return mod
