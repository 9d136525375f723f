"""diffusioniqt_amd — MI355X-native engine for the DiffusionIQT hot path (3-D conditional diffusion
image-quality transfer).  Device math lives in csrc/ (HIP, gfx950) behind the C ABI of include/diqt.h."""
__version__ = "0.1.0"
