#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Regenerates the golden fixtures from /root/reference into a scratch directory (oracle/gen_golden.py with
MG_GOLDEN_OUT set) and compares every array with the committed tests/golden/*.npz: `python oracle/check_golden.py [group ...]`,
groups as gen_golden.py's argument ("" = the round-1 cases, time_model, round4, round5, round5b).  Exit code 0 = every array of
every regenerated file equal to the committed one (the archives themselves carry time stamps)."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")


def check(groups):
    bad, seen = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for group in groups:
            cmd = [sys.executable, os.path.join(HERE, "gen_golden.py")] + ([group] if group else [])
            subprocess.run(cmd, check=True, env=dict(os.environ, MG_GOLDEN_OUT=tmp), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for name in sorted(os.listdir(tmp)):
            new, old = np.load(os.path.join(tmp, name)), np.load(os.path.join(GOLDEN, name))
            if sorted(new.files) != sorted(old.files):
                bad.append((name, "keys differ"))
                continue
            for key in new.files:
                a, b = new[key], old[key]
                if a.shape != b.shape or a.dtype != b.dtype or not (np.array_equal(a, b, equal_nan=True) if a.dtype.kind in "fc" else np.array_equal(a, b)):
                    bad.append((name, key))
            seen += 1
    return seen, bad


if __name__ == "__main__":
    groups = sys.argv[1:] or ["", "time_model", "round4", "round5", "round5b"]
    seen, bad = check(groups)
    print("%d fixture(s) regenerated, %d array(s) differ" % (seen, len(bad)))
    for item in bad:
        print("  ", *item)
    sys.exit(1 if bad else 0)
