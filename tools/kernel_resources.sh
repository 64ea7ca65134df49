#!/bin/bash
# VGPRs / spills / occupancy / LDS of every kernel of one csrc file (no GPU needed):
#   tools/kernel_resources.sh erosion_particles_tiled.hip [grep -E pattern on the demangled name] [extra hipcc flags...]
f=${1:-erosion_particles_tiled.hip}; pat=${2:-.}; shift 2 2>/dev/null
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -fno-fast-math \
  -munsafe-fp-atomics -fno-gpu-rdc "$@" -c /root/repo/soillib_amd/csrc/$f -o /tmp/kr_$$.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | sed 's/ *\[-Rpass.*//' |
  awk '/Function Name:/ {name=$NF} / VGPRs:/ {v=$NF} /AGPRs:/ {a=$NF} /ScratchSize/ {sc=$NF} /VGPRs Spill/ {sp=$NF} /SGPRs Spill/ {ss=$NF} /Occupancy/ {oc=$NF} /LDS Size/ {print v, a, sp, ss, sc, oc, $NF, name}' |
  while read v a sp ss sc oc lds name; do echo "vgpr=$v agpr=$a vspill=$sp sspill=$ss scratch=$sc occ=$oc lds=$lds $(echo $name | c++filt | sed 's/(.*//')"; done | grep -E "$pat"
rm -f /tmp/kr_$$.o
