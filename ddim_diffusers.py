#!/usr/bin/env python
"""Drop-in entry point with the reference's script name (``accelerate launch ddim_diffusers.py ...`` or
plain ``python``).  See bndm_amd/cli_ddim.py."""
import sys

from bndm_amd.cli_ddim import main

if __name__ == "__main__":
    sys.exit(main())
