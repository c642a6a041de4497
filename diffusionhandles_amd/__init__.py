"""diffusionhandles_amd -- MI355X-native guided-denoising edit path with the DiffusionHandles API.

The compute path is the HIP library (libdiffhandles_hip.so, C ABI in include/diffhandles_hip.h);
importing this package does not need a GPU, calling into it does.
"""
from .diffusion_handles import DiffusionHandles  # noqa: F401
from .guided_stable_diffuser import GuidedStableDiffuser, StepGuidanceWeightSchedule  # noqa: F401
from .stable_null_inverter import StableNullInverter  # noqa: F401

__all__ = ["DiffusionHandles", "GuidedStableDiffuser", "StableNullInverter", "StepGuidanceWeightSchedule"]
