"""Which kernels of the shipped library ran on the simulator: merges the coverage files of one or more tools/sim_suite.sh runs.
    python tools/sim_coverage.py <work dir> <out.txt> <suite log> [<suite log> ...]      (the PASS lines of the logs name the configurations)"""
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    from tests.gfx950sim import suite
    from tests.hipmock import harness as H
    work, out, logs = sys.argv[1], sys.argv[2], sys.argv[3:]
    names = []
    for lg in logs:
        for ln in open(lg):
            m = re.match(r"PASS\s+(\S+)", ln)
            if m and m.group(1) not in names and os.path.exists(os.path.join(work, f"coverage_{m.group(1)}.txt")):
                names.append(m.group(1))
    suite.write_coverage(out, os.path.abspath(H.PRODUCT_LIB), suite.coverage_from_work(work, names), names)
    print(open(out).read().split("\n")[2])


if __name__ == "__main__":
    main()
