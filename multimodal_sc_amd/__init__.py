"""multimodal_sc_amd — MI355X-native (gfx950) multimodal semantic-communication hot path.

Host side: PyTorch-ROCm for device memory / streams / torch.distributed only.  Compute: hand-written HIP kernels
behind the C-ABI in include/msc.h (multimodal_sc_amd/csrc/libmsc_hip.so).  There is no CPU or eager fallback.
"""
from .model import MSCConfig, MultimodalSC, named_config  # noqa: F401
from .runtime import ParamStore, Runtime  # noqa: F401
