#!/bin/bash
# tools/build_variant.sh <name> <file.hip> "<-D flags>": libmfhip_<name>.so = the product objects with ONE source file
# rebuilt under extra flags (tuning A/B on the GPU box: MF_LIBMFHIP=libmfhip_<name>.so selects it)
set -e
cd "$(dirname "$0")/../morefusion_amd/csrc"
name=$1; src=$2; flags=$3
make -s >/dev/null
mkdir -p _obj_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -Wno-pass-failed $flags -c $src -o _obj_$name/${src%.hip}.o
objs=""
for o in _obj/*.o; do b=$(basename $o); if [ "$b" = "${src%.hip}.o" ]; then objs="$objs _obj_$name/$b"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmfhip_$name.so $objs
echo built libmfhip_$name.so
