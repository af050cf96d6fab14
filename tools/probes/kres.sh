#!/bin/bash
# usage: tools/probes/kres.sh file.hip  -> compact per-kernel resource table (VGPR / scratch / occupancy)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | \
awk '/error|warning:/ {print} /Function Name:/ {name=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /Occupancy/ {o=$(NF-1)} /LDS Size/ {printf "v=%s a=%s scr=%s occ=%s lds=%s %s\n", v, a, s, o, $(NF-1), name}' | c++filt | sed 's/(anonymous namespace):://g' | cut -c1-110
