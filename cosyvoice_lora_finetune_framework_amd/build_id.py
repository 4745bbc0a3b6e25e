"""Identity of the kernel sources a libcvft.so build comes from: the first 16 hex digits of a SHA-256 over csrc/*.hip, csrc/*.h and
include/cvft.h (names and bytes, sorted).  Every profiles/*.json that bench.py quotes (counter traffic, MFMA-busy, launches per
step) records it when it is written (tools/pmc_traffic.py, tools/counters.py, tools/step_summary.py); bench.py prints null plus
the reason instead of a figure whose recorded identity is not the running tree's -- a kernel commit without
tools/refresh_profiles.sh can no longer make the bench line quote stale counters.  No torch import: tools use it too."""
import glob
import hashlib
import os

_PKG = os.path.dirname(os.path.abspath(__file__))


def csrc_sha16() -> str:
    files = sorted(glob.glob(os.path.join(_PKG, "csrc", "*.hip")) + glob.glob(os.path.join(_PKG, "csrc", "*.h")))
    files.append(os.path.join(os.path.dirname(_PKG), "include", "cvft.h"))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_sha16())
