"""Flag surface of the reference's input_args.py (parse_args used by latent_iadb_bn_diffusers.py:43):
forwards to the MI355X CLI parser (BNDM flags honoured, HF training flags accepted and ignored)."""
from bndm_amd.cli_latent import build_parser


def parse_args(argv=None):
    return build_parser().parse_args(argv)
