#!/bin/bash
# Register / LDS / spill report of the fused vector kernel's instances (device-only compile of one format's
# translation unit with the library's flags): scripts/kernel_resources.sh [fmt 0..3] [extra -D flags] | grep 'ILi4ELi3E'
fmt=${1:-0}; shift
cd "$(dirname "$0")/../gpuacceleratedtracking_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -DGAT_BUILD -I../../include \
  --offload-device-only -c gat_dc_f${fmt}.hip -o /tmp/gat_dc_f${fmt}.co -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
  sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' |
  awk '/Function Name/ {name=$NF} /VGPRs:/ {v=$NF} /AGPRs:/ {ag=$NF} /SGPRs:/ {s=$NF} /Spill/ {sp=sp" "$NF} /ScratchSize/ {sc=$NF} /Occupancy/ {o=$NF} /LDS Size/ {print name, "vgpr", v, "agpr", ag, "sgpr", s, "spill", sp, "scratch", sc, "occ", o; sp=""}' | c++filt | sed 's/void gat:://; s/(gat::DcArgs)//'
