"""Drop-in module path of the reference (``from bluenoise.get_noise_recent import get_noise_v2``,
iadb_bn.py:26, latent_iadb_bn_diffusers.py:39, gradio_bndm.py:8): forwards to the MI355X path."""
from bndm_amd.bluenoise import get_noise, get_noise_v2, noise_padding  # noqa: F401
