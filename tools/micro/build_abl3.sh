#!/bin/bash
# Dev: build k_spconv3 variants (IRX_S3_ABL / IRX_S3_AMAP / other -D flags in csrc/irx_spconv3.hip) into
# tools/micro/libirx_s3_<tag>.so.  Usage: build_abl3.sh tag1:"-DIRX_S3_ABL=1" tag2:"-DIRX_S3_AMAP=1" ...
set -e
cd "$(dirname "$0")/../../instancerefer_amd/csrc"
O=/tmp/abl3; mkdir -p $O
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast"
for f in *.hip; do
  [ $f = irx_spconv3.hip ] && continue
  extra=""; case $f in irx_labels.hip|irx_project.hip) extra="-ffp-contract=off";; esac
  { [ -f $O/$f.o ] && [ $O/$f.o -nt $f ] && [ $O/$f.o -nt irx_common.h ]; } || hipcc $FL $extra -c $f -o $O/$f.o &
done
for spec in "$@"; do tag=${spec%%:*}; fl=${spec#*:}; hipcc $FL $fl -c irx_spconv3.hip -o $O/s3_$tag.obj & done
wait
for spec in "$@"; do tag=${spec%%:*}; hipcc --offload-arch=gfx950 -shared -fPIC $(ls $O/*.hip.o) $O/s3_$tag.obj -o ../../tools/devlib/libirx_s3_$tag.so; done
ls -la ../../tools/devlib/libirx_s3_*.so
