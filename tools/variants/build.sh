#!/bin/bash
# usage: tools/variants/build.sh name:"-Dflags" ...   builds tools/variants/lib_<name>.so from the working tree (in parallel)
ROOT=$(cd $(dirname $0)/../.. && pwd)
for v in "$@"; do n=${v%%:*}; f=${v#*:}
  (cd $ROOT/celeste.jl_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wl,--version-script=exports.map $f -o $ROOT/tools/variants/lib_$n.so celeste_abi.hip -L/opt/rocm/lib -lrccl -lpthread 2>/dev/null) &
done
wait
ls $ROOT/tools/variants/*.so
