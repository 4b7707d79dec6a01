#!/bin/bash
# Per-kernel register / scratch / LDS / occupancy report of the HIP sources (hipcc -Rpass-analysis=kernel-resource-usage),
# one line per kernel instantiation.  usage: tools/resource_usage.sh [file.hip ...]   (default: every kernel file)
cd "$(dirname "$0")/../probabilisticsemslam_amd/csrc" || exit 1
files=("$@")
[ ${#files[@]} -eq 0 ] && files=(kbest_engine.hip kbest_lane.hip kbest_small.hip kbest_tiny.hip kbest_bnb.hip kbest_wide.hip kbest_exact.hip kbest_merge.hip kbest_costs.hip)
for f in "${files[@]}"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. \
      -c -o /dev/null -x hip "$f" -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk -v file="$f" '
    /Function Name:/ { name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name) }
    /TotalSGPRs:/ { sg=$0; sub(/.*TotalSGPRs: /,"",sg); sub(/ .*/,"",sg) }
    / VGPRs:/ { vg=$0; sub(/.* VGPRs: /,"",vg); sub(/ .*/,"",vg) }
    /AGPRs:/ { ag=$0; sub(/.*AGPRs: /,"",ag); sub(/ .*/,"",ag) }
    /ScratchSize \[bytes\/lane\]:/ { sc=$0; sub(/.*: /,"",sc); sub(/ .*/,"",sc) }
    /Occupancy \[waves\/SIMD\]:/ { oc=$0; sub(/.*: /,"",oc); sub(/ .*/,"",oc) }
    /SGPRs Spill:/ { ss=$0; sub(/.*: /,"",ss); sub(/ .*/,"",ss) }
    /VGPRs Spill:/ { vs=$0; sub(/.*: /,"",vs); sub(/ .*/,"",vs) }
    /LDS Size \[bytes\/block\]:/ { l=$0; sub(/.*: /,"",l); sub(/ .*/,"",l);
        cmd="echo " name " | c++filt"; cmd | getline dn; close(cmd); sub(/\(kb::.*/,"",dn);
        printf "%-16s %-58s VGPR %3s AGPR %s SGPR %3s spillV %3s spillS %3s scratch %4s B/lane occ %s LDS(static) %s\n", file, dn, vg, ag, sg, vs, ss, sc, oc, l }'
done
