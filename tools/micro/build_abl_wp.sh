#!/bin/bash
# Dev: k_wgrad_pairs ablation variants (IRX_WP_ABL in csrc/irx_pairs.hip) -> tools/micro/libirx_wp<mask>.so
set -e
cd "$(dirname "$0")/../../instancerefer_amd/csrc"
O=/tmp/ablwp; mkdir -p $O
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast"
for f in *.hip; do
  [ $f = irx_pairs.hip ] && continue
  extra=""; case $f in irx_labels.hip|irx_project.hip) extra="-ffp-contract=off";; esac
  hipcc $FL $extra -c $f -o $O/$f.o 2>/dev/null &
done
for m in "$@"; do hipcc $FL -DIRX_WP_ABL=$m -c irx_pairs.hip -o $O/wp_$m.obj 2>/dev/null & done
wait
for m in "$@"; do hipcc --offload-arch=gfx950 -shared -fPIC $(ls $O/*.hip.o) $O/wp_$m.obj -o ../../tools/micro/libirx_wp$m.so; done
ls ../../tools/micro/libirx_wp*.so
