#!/bin/bash
# usage: tools/isa.sh diart_amd/csrc/k_xxx.hip [outdir]   -> per-kernel register / LDS summary, .s kept in outdir
SRC=$1; OUT=${2:-/tmp/isa}; mkdir -p $OUT
cd $OUT && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -x hip -c $OLDPWD/$SRC -o $OUT/tmp.o -save-temps=obj \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy" |
  sed -e 's/.*remark: [^ ]* *//' -e 's/\[-Rpass.*//' | paste - - - - - | sed -e 's/Function Name: //' | c++filt | cut -c1-200
