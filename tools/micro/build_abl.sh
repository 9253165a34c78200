#!/bin/bash
# Dev: build k_spconv2 ablation variants (see IRX_S2_ABL in csrc/irx_spconv2.hip) into tools/micro/libirx_abl<mask>.so
set -e
cd "$(dirname "$0")/../../instancerefer_amd/csrc"
O=/tmp/abl; mkdir -p $O
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast"
for f in *.hip; do
  [ $f = irx_spconv2.hip ] && continue
  extra=""; case $f in irx_labels.hip|irx_project.hip) extra="-ffp-contract=off";; esac
  [ -f $O/$f.o ] && [ $O/$f.o -nt $f ] || hipcc $FL $extra -c $f -o $O/$f.o &
done
for m in "$@"; do hipcc $FL -DIRX_S2_ABL=$m -c irx_spconv2.hip -o $O/s2_$m.o & done
wait
for m in "$@"; do hipcc --offload-arch=gfx950 -shared -fPIC $(ls $O/*.hip.o) $O/s2_$m.o -o ../../tools/micro/libirx_abl$m.so; done
ls -la ../../tools/micro/libirx_abl*.so
