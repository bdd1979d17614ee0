#!/bin/bash
# Per-kernel register / scratch / occupancy report of one or more kernel translation units (build container, no GPU needed):
#   tools/kernel_resources.sh dtof_shade_plain dtof_shade_res0 [-- extra hipcc flags]   ->  one line per kernel on stdout
# (clang's -Rpass-analysis=kernel-resource-usage remarks, folded to one line per kernel)
cd "$(dirname "$0")/../mitsuba3dopplertof_amd/csrc" || exit 1
tus=(); extra=()
while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; extra=("$@"); break; fi; tus+=("$1"); shift; done
for tu in "${tus[@]}"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize "${extra[@]}" \
      -Rpass-analysis=kernel-resource-usage --cuda-device-only -c "$tu.hip" -o /dev/null 2>&1 | sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' |
  awk -v tu="$tu" '
    /Function Name:/ { name=$NF }
    / VGPRs:/ { v=$NF } /AGPRs:/ { a=$NF } /TotalSGPRs:/ { sg=$NF } /SGPRs Spill:/ { ss=$NF } /VGPRs Spill:/ { vs=$NF }
    /ScratchSize/ { sc=$NF } /Occupancy/ { oc=$NF }
    /LDS Size/ { lds=$NF; cmd="c++filt " name; cmd | getline dem; close(cmd); sub(/\(dtof::ShadeArgs\)/,"",dem); sub(/void dtof::/,"",dem);
                 printf "%-18s %-46s vgpr %3s agpr %3s sgpr %3s | spilled v %3s s %3s | scratch %4s B/lane | waves/SIMD %s | static lds %s\n", tu, dem, v, a, sg, vs, ss, sc, oc, lds }'
done
