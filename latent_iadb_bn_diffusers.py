#!/usr/bin/env python
"""Drop-in entry point with the reference's script name.  See bndm_amd/cli_latent.py."""
import sys

from bndm_amd.cli_latent import main

if __name__ == "__main__":
    sys.exit(main())
