"""Drop-in alias: `from diffhandles import DiffusionHandles` resolves to the MI355X-native package."""
from diffusionhandles_amd import DiffusionHandles  # noqa: F401
