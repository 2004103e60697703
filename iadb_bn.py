#!/usr/bin/env python
"""Drop-in entry point with the reference's script name: ``python iadb_bn.py --dataset=... --train_or_test=test``
(scripts/sampling/*.sh).  See bndm_amd/cli_iadb.py."""
import sys

from bndm_amd.cli_iadb import main

if __name__ == "__main__":
    sys.exit(main())
