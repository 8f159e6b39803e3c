"""adt_str_amd -- MI355X-native hot path of pier-maker92/ADT_STR.

Host side (Python on PyTorch-ROCm) of the hand-written gfx950 kernels in
``csrc/`` (built into ``libadt_hip.so``, C ABI declared in ``include/adt_hip.h``).
There is no CPU fallback: every op raises if the library or a GPU is missing.
"""
__version__ = "0.1.0"
