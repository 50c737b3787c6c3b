#!/bin/bash
# Round 5: the same-box A/B series behind DESIGN.md section 4.1 / 9 (outputs: gpurun_out/r05/ab_<series>.txt; judged copies in
# profiles/r05/).  One series per call:  scripts/r05_ab.sh SERIES       (run through gpurun; builds are made on the CPU first)
# Variant libraries: python -m gpuacceleratedtracking_amd.build --variant NAME -DGAT_DC_DEV [-D...]  -> build/libgat_NAME.so
# (development builds: only the BASELINE shapes' instances; `base` = the current text without extra flags).
#   ablate   base + -DGAT_DC_ABLATE={1,2,4,8,12,16,13,29} as abl1 ... abl29      what each part of the kernel costs (c2, i8)
#   rule     base                                                                 where the two-channel 2 x 2 tile pays
#   quads    base                                                                 quads / sign-bit tables on the two-channel tile
#   k32      (product library)                                                    configs[3] as a whole on one GPU: tilings
# Series of states that no longer build from this tree (the one-channel 2 x 2 tile, three-wave and one-sample-pass forms of the
# two-channel tile, the two-plane replica layout, int8 quads) are recorded in profiles/r05/ab_*_not_kept.txt / ab_kt2.txt; the
# two-plane layout is scripts/experiments/r05_two_plane_replica.patch.
set -o pipefail
mkdir -p gpurun_out/r05
series=$1; out=gpurun_out/r05/ab_$series.txt; : > $out
q() { tag=$1; lib=$2; opts=$3; shift 3; QARGS="$opts" GAT_LIBRARY=${lib:+$PWD/$lib} bash scripts/r05_quick.sh $tag "$@" | tee -a $out; }
B=build/libgat_base.so
case $series in
ablate)
  L="base:$B"; for n in 1 2 4 8 12 16 13 29; do L="$L abl$n:build/libgat_abl$n.so"; done
  bash scripts/r05_ablate.sh r05/ablate_c2.txt c2 $L; bash scripts/r05_ablate.sh r05/ablate_i8.txt i8 $L ;;
rule)
  SH="c1k2 c1k3 c1k4 c1k5 c1k7 i16k8 ilk8 m8k4 m12k4 c2i16 lat12 lat4"
  for rep in 1 2; do q one_channel $B "--option dc_aw2=0" $SH; q two_channel $B "--option dc_aw2=1" $SH; done ;;
quads)
  for rep in 1 2; do
    q one_channel $B "--option dc_aw2=0" c2 c2l1 c1k8
    q two_channel_per_entry $B "--option dc_aw2=1 --option dc_quads=0" c2 c2l1 c1k8
    q two_channel_quads $B "--option dc_aw2=1" c2 c2l1 c1k8
    q two_channel_quads_int8_tables $B "--option dc_aw2=1 --option dc_bits=0" c2
    q two_channel_quads_bit_tables $B "--option dc_aw2=1 --option dc_bits=2" c2l1
  done ;;
k32)
  for rep in 1 2; do
    q kt4 "" "" c3k32; q kt2 "" "--option dc_kt=2" c3k32; q kt1 "" "--option dc_kt=1" c3k32
    q split_bf16 "" "--matrix-core 3" c3k32; q two_by_two "" "--option dc_aw=2 --option dc_aw2=1" c3k32
  done ;;
*) echo "series: ablate rule quads k32"; exit 2 ;;
esac
