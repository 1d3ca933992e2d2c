#!/bin/bash
# prints VGPR/SGPR/occupancy of every kernel in kyhip.hip (extra flags: $@)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -no-hip-rt -Rpass-analysis=kernel-resource-usage "$@" -o /tmp/kyhip_res.so ky_amd/csrc/kyhip.hip 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs:|SGPRs:|Occupancy|ScratchSize|LDS Size|Spill" | sed 's/.*remark: *//; s/ \[-Rpass.*//' \
 | awk '/Function Name/{if(line)print line; line=$0; next}{line=line" | "$0}END{print line}' | sed 's/Function Name: //'
