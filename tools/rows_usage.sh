#!/bin/bash
# resource usage of one instantiation of the row-layout local-energy kernel: tools/rows_usage.sh N SPLIT [extra flags]
cd /root/repo/fermiflow_amd/csrc
N=${1:-6}; S=${2:-1}; shift; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast --cuda-device-only -DFF_TN=$N -DFF_TS=$S "$@" -c _rows_probe.hip -o /tmp/rows_probe.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "VGPRs:|AGPRs|Scratch|LDS Size|SGPRs:|Occupancy" | sed 's/.*remark: [^ ]* *//; s/\[-Rpass.*//' | paste - - - - - - -
