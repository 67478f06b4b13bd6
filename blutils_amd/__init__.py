"""blutils_amd — MI355X-native engine for blutils' per-query taxonomic consensus.

The compute path is libblu_consensus.so (C ABI in include/blu_consensus.h, HIP
kernels for gfx950).  Python here is a thin host-side harness: ctypes bindings,
torch tensors as device memory, synthetic workloads.  There is no CPU fallback:
importing `blutils_amd.engine` without the built library raises.
"""
__version__ = "0.1.0"
