"""wavjepa_amd: MI355X-native (gfx950) WavJEPA pre-training step behind the reference's Python surface.

Sub-modules are imported lazily by users (`from wavjepa_amd.jepa import JEPA`); importing the package itself never
touches the GPU.  The compute path lives in `csrc/` (HIP) behind the C ABI of `include/wavjepa_hip.h`.
"""
__version__ = "0.1.0"
