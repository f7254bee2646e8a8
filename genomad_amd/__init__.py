"""genomad_amd — MI355X-native nn-classification hot path of geNomad.

Host side is plain Python + numpy over a ctypes C-ABI (include/genomad_nn.h);
the arithmetic runs in hand-written HIP kernels for gfx950 (genomad_amd/csrc).
No torch / tensorflow import happens here.
"""
__version__ = "0.1.0"
