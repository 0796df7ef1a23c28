#!/bin/bash
# kernels of one translation unit that use scratch (private memory): tools/scratch_report.sh <file.hip>
# (a register array indexed by a run-time value - `pr[d]` under `for (d < g.dim)` - lands there; every access is a
# global-memory round trip and the dispatch has to set the scratch wave state up)
F=${1:?file.hip}
cd /tmp && /opt/rocm/bin/hipcc -c "$F" -o /tmp/scratch_report.o -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off \
  -fno-slp-vectorize --cuda-device-only -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "Function Name|ScratchSize|  VGPRs:" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | paste - - - |
  sed -e 's/Function Name: //' -e 's/ScratchSize \[bytes\/lane\]: /scratch=/' | awk '{print $NF, $2, $3, $1}' | sort -u | grep -v "^scratch=0 "
