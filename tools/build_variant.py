"""Build a compile-time variant of the HIP library next to the product build, for A/B timing on the GPU box:
    python tools/build_variant.py NAME [-DFLAG ...]      ->  pcl_tracking_amd/_build/var_NAME.so
    PFT_LIB_PATH=pcl_tracking_amd/_build/var_NAME.so python bench.py ...
(the variants travel with the snapshot like the product build; hipcc cross-compiles here, so no GPU time is spent compiling)"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcl_tracking_amd import build as B  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(B.OUT_DIR, "var_%s.so" % name)
os.makedirs(B.OUT_DIR, exist_ok=True)
cmd = [B.hipcc()] + B.HIPCC_FLAGS + flags + ["-o", out] + [os.path.join(B.CSRC, s) for s in B.SOURCES]
r = subprocess.run(cmd, capture_output=True, text=True)
if r.returncode != 0:
    sys.stderr.write(r.stdout + r.stderr)
    sys.exit(1)
print(out)
