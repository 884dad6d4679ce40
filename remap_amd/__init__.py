"""remap_amd — MI355X-native alignment hot path of kataklinger/remap (kpe / kpm / frc / fgm / fgs).

The product is the C-ABI library built from ``remap_amd/csrc`` (``include/remap_amd.h``) and
the C++ header surface under ``include/remap``; this package is the thin Python host side
used by tests and bench: a ctypes binding (``remap_amd.lib``), the batched pipeline driver
(``remap_amd.pipeline``) and the synthetic stream generators (``remap_amd.synth``).
"""
__version__ = "0.1.0"
