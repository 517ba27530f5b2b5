#!/bin/bash
# Build another copy of the library with extra compiler flags (diagnostic / tuning variants), e.g.
#   tools/build_variant.sh tune -DUBR_TUNE        -> ubresnet_amd/libubr_tune.so   (load with UBR_LIB=...)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
T=$(mktemp -d)
objs=""
for f in $R/ubresnet_amd/csrc/*.hip; do
  o=$T/$(basename ${f%.hip}).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-gpu-rdc -ffp-contract=off "$@" -c $f -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ubresnet_amd/libubr_$NAME.so $objs
rm -rf $T
echo "built ubresnet_amd/libubr_$NAME.so ($*)"
