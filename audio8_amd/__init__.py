"""audio8_amd -- MI355X-native (gfx950) wav2vec2 hot path for mead-ml/audio8.

Host side: Python on PyTorch-ROCm (device memory, streams, torch.distributed only).
Arithmetic: hand-written HIP kernels in liba8hip.so behind the C ABI of include/a8hip.h.
"""
__version__ = "0.1.0"
