"""Drop-in module path of the reference's utils.py (imported by gradio_bndm.py:6 as
``from utils import get_model, sample_iadb, ...``): forwards to the MI355X path."""
from bndm_amd.sampler import get_model, sample_iadb, sample_iadb_conditional  # noqa: F401
from bndm_amd.schedules import get_scheduler, get_scheduler_gamma  # noqa: F401
