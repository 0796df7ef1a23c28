#!/bin/bash
# Build a VARIANT of liblbhip.so with extra compiler flags into tools/bin/ (A/B runs on one box: tools/ab_lib.sh):
#   tools/build_variant.sh <tag> <extra flags...>      e.g.  tools/build_variant.sh i0 -DLB_GEMM_INTERLEAVE=0
set -e
cd "$(dirname "$0")/.."
TAG=$1; shift
OUT=tools/bin/var_$TAG
mkdir -p $OUT
python - "$OUT" "$@" <<'PY'
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.getcwd())
from lagrangebench_amd import build as B
out, extra = sys.argv[1], sys.argv[2:]
hipcc = B._hipcc()
objs = []
def one(src):
    o = os.path.join(out, src.replace(".hip", ".o"))
    cmd = [hipcc, "-c", os.path.join(B.CSRC, src), "-o", o] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + extra
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return o
with ThreadPoolExecutor(max_workers=4) as ex:
    objs = list(ex.map(one, B.SOURCES))
libdir = os.path.join(B._rocm_root(), "lib")
subprocess.check_call([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", os.path.join(out, "liblbhip.so")] + objs +
                      ["-ldl", "-Wl,-rpath," + libdir])
print("built", os.path.join(out, "liblbhip.so"))
PY
