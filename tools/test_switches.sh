#!/bin/bash
# The parity subset under every surviving kernel / schedule switch now runs as ordinary GPU tests:
#   python -m pytest tests/test_switches_gpu.py -m gpu -q
# (round 3 retired the round-1 / rejected kernel variants and their switches from liblbhip.so; round 4 removed their sources)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
exec python -m pytest tests/test_switches_gpu.py -m gpu -q "$@"
