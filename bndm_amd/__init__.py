"""bndm_amd -- MI355X-native implementation of the bndm sampling hot path.

Python keeps the reference's call signatures (bluenoise.get_noise_recent.get_noise_v2,
utils.sample_iadb / get_model / get_scheduler*, the scripts/sampling CLIs) and forwards to
hand-written gfx950 HIP kernels in libbndm_hip.so through the C ABI in include/bndm_hip.h.
"""
__version__ = "0.1.0"
