"""sha256 over the kernel sources exactly as multimodal_sc_amd/csrc/Makefile embeds it (BUILD_ID): `cat $(sort *.hip) msc_common.h
gemm_plan.h ../../include/msc.h | sha256sum | cut -c1-16` (+ the other headers the Makefile lists in HDRS)."""
import glob
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multimodal_sc_amd", "csrc")


def source_build_id():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    hdrs = re.search(r"^HDRS\s*=\s*(.*)$", mk, flags=re.M).group(1).split()
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + [os.path.normpath(os.path.join(CSRC, h)) for h in hdrs]
    h = hashlib.sha256()
    for f in files:
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
