#!/bin/bash
# Dev: build k_wgrad3 variants (IRX_W3_ABL and other -D flags in csrc/irx_pairs.hip) into
# tools/micro/libirx_w3_<tag>.so.  Usage: build_ablw3.sh tag1:"-DIRX_W3_ABL=1" tag2:"-DIRX_W3_ABL=2" ...
set -e
cd "$(dirname "$0")/../../instancerefer_amd/csrc"
O=/tmp/ablw3; mkdir -p $O
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast"
for f in *.hip; do
  [ $f = irx_pairs.hip ] && continue
  extra=""; case $f in irx_labels.hip|irx_project.hip) extra="-ffp-contract=off";; esac
  { [ -f $O/$f.o ] && [ $O/$f.o -nt $f ] && [ $O/$f.o -nt irx_common.h ]; } || hipcc $FL $extra -c $f -o $O/$f.o &
done
for spec in "$@"; do tag=${spec%%:*}; fl=${spec#*:}; hipcc $FL $fl -c irx_pairs.hip -o $O/w3_$tag.obj & done
wait
for spec in "$@"; do tag=${spec%%:*}; hipcc --offload-arch=gfx950 -shared -fPIC $(ls $O/*.hip.o) $O/w3_$tag.obj -o ../../tools/micro/libirx_w3_$tag.so; done
ls -la ../../tools/micro/libirx_w3_*.so
