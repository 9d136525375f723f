#!/bin/bash
# Builds libdiqt_hip.so (gfx950) next to this script.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-result"
OBJS=""
PIDS=""
for f in conv_mfma conv_half elementwise bgemm conv_direct attention datapath; do
  if [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ common.h -nt $f.o ] || [ ../../include/diqt.h -nt $f.o ]; then
    rm -f $f.o
    $HIPCC $FLAGS -c $f.hip -o $f.o &
    PIDS="$PIDS $!"
  fi
  OBJS="$OBJS $f.o"
done
if [ ! -f lib.o ] || [ lib.cpp -nt lib.o ] || [ common.h -nt lib.o ]; then
  rm -f lib.o
  $HIPCC $FLAGS -x hip -c lib.cpp -o lib.o &
  PIDS="$PIDS $!"
fi
for p in $PIDS; do wait $p || { echo "build FAILED"; exit 1; }; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libdiqt_hip.so $OBJS lib.o
echo "built $(pwd)/libdiqt_hip.so"
