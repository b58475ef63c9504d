#!/bin/bash
# register use of the quad kernel's two main variants for (segments per quad, waves per SIMD) pairs: tools/regs.sh "2 4" "3 4"
cd "$(dirname "$0")/../geneticscre_amd/csrc"
for v in "$@"; do set -- $v; printf "QSEGS=%s QWAVES=%s: " $1 $2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -DGCRE_IEQ_ONLY -DGCRE_QSEGS=$1 -DGCRE_QWAVES=$2 $EXTRA -Rpass-analysis=kernel-resource-usage -c gcre_ieq.hip -o /tmp/ieq.o 2>&1 | grep -E "error|warning:|  VGPRs:|ScratchSize|Spill|Occupancy" | sed 's/.*remark: *//; s/\[-Rpass.*//' | tr '\n' ' ' | sed 's/  */ /g'; echo; done
