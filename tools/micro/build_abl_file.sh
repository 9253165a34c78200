#!/bin/bash
# Dev: build variants of ONE source file with extra -D flags into tools/micro/libirx_<prefix>_<tag>.so.
# Usage: build_abl_file.sh irx_norm.hip bn u8:"-DBN_U=8" u16:"-DBN_U=16" ...
set -e
SRC=$1; PFX=$2; shift 2
cd "$(dirname "$0")/../../instancerefer_amd/csrc"
O=/tmp/abl_$PFX; mkdir -p $O
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast"
for f in *.hip; do
  [ $f = $SRC ] && continue
  extra=""; case $f in irx_labels.hip|irx_project.hip) extra="-ffp-contract=off";; esac
  { [ -f $O/$f.o ] && [ $O/$f.o -nt $f ] && [ $O/$f.o -nt irx_common.h ]; } || hipcc $FL $extra -c $f -o $O/$f.o &
done
for spec in "$@"; do tag=${spec%%:*}; fl=${spec#*:}; hipcc $FL $fl -c $SRC -o $O/v_$tag.obj & done
wait
for spec in "$@"; do tag=${spec%%:*}; hipcc --offload-arch=gfx950 -shared -fPIC $(ls $O/*.hip.o) $O/v_$tag.obj -o ../../tools/micro/libirx_${PFX}_$tag.so; done
ls -la ../../tools/micro/libirx_${PFX}_*.so
