"""MI355X-native burst super-resolution hot path — a drop-in for the reference package of the same
name (Jamy-L/Handheld-Multi-Frame-Super-Resolution): ``from handheld_super_resolution import process``.

Host code is Python on PyTorch-ROCm tensors; every stage is a hand-written HIP kernel for gfx950
reached through the C ABI of libhhsr_hip.so (include/hhsr.h).  Importing the package does not need a
GPU; calling into the hot path without one (or without the built library) raises RuntimeError."""
from .config import Config, OmegaConf, default_config  # noqa: F401
from .super_resolution import process, main, prepare_config, BurstPipeline  # noqa: F401

__all__ = ["process", "main", "prepare_config", "BurstPipeline", "Config", "OmegaConf", "default_config"]
