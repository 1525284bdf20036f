#!/bin/bash
# resource usage of one instantiation of the matrix-core local-energy kernel: tools/mfma_usage.sh N WPS [extra flags]
cd /root/repo/fermiflow_amd/csrc
N=${1:-6}; W=${2:-1}; shift; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast --cuda-device-only -DFF_TN=$N -DFF_TW=$W "$@" -c _mfma_probe.hip -o /tmp/mfma_probe.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "VGPRs:|AGPRs|Scratch|LDS Size|Occupancy" | sed 's/.*remark: [^ ]* *//; s/\[-Rpass.*//' | paste - - - - - | tail -1
