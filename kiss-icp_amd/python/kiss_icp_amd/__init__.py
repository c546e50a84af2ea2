"""kiss_icp_amd -- MI355X-native KISS-ICP registration hot path behind the reference's interfaces.

Everything that computes goes through libkicp.so (hand-written HIP kernels for gfx950, C-ABI in
include/kicp.h).  Importing this package does not load the library; the first call does, and it
fails loudly if the library or the GPU is missing -- there is no CPU fallback.
"""
__version__ = "0.1.0"
