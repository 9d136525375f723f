#!/bin/bash
# Builds libdiqt_hip.so (gfx950) next to this script.  hipcc cross-compiles without a GPU.
#   build.sh           recompiles a translation unit when the SHA-256 of (its source, the local headers it includes, the flags)
#                      differs from the stamp written beside its object file (no reliance on mtimes)
#   build.sh --force   deletes every object / stamp / library first: a provably clean build (what __graft_entry__.build() runs)
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-result"
if [ "$1" = "--force" ]; then
  rm -f *.o *.stamp libdiqt_hip.so
fi
# the local headers a source includes, transitively (quoted includes only; they all live beside the sources or in ../../include)
deps_of() {
  local seen="" todo="$1" f inc
  while [ -n "$todo" ]; do
    set -- $todo; f=$1; shift; todo="$*"
    case " $seen " in *" $f "*) continue;; esac
    seen="$seen $f"
    for inc in $(sed -n 's/^[[:space:]]*#include[[:space:]]*"\([^"]*\)".*/\1/p' "$f"); do
      [ -f "$inc" ] && todo="$todo $inc"
    done
  done
  echo $seen
}
stamp_of() { cat $(deps_of "$1") | cat - <(echo "$FLAGS $HIPCC $2") | sha256sum | cut -d' ' -f1; }
OBJS=""
PIDS=""
BUILT=""
for f in conv_fwd9 conv_fwd9_d conv_fwd9_b conv_fwd9_c conv_fwd9_e conv_fwd9_f conv_fwd9_g conv_fwd9_h conv_fwd9_i conv_mfma conv_wgrad conv_wgrad_h conv_pw conv_half conv_f9h conv_f9h_b conv_f9h_c conv_smallcout elementwise bgemm conv_direct attention attn_temporal datapath lib; do
  src=$f.hip; xflag=""
  if [ $f = lib ]; then src=lib.cpp; xflag="-x hip"; fi
  # the GroupNorm-apply instantiations of conv_fwd9_kernel: their fully unrolled step loop exceeds LLVM's default size limit for
  # `#pragma unroll` (16384), below which the plain instantiations stay
  case $f in conv_fwd9_f|conv_fwd9_g|conv_fwd9_h|conv_fwd9_i) xflag="-mllvm -pragma-unroll-threshold=1000000";; esac
  want=$(stamp_of $src "$xflag")
  if [ ! -f $f.o ] || [ ! -f $f.stamp ] || [ "$(cat $f.stamp)" != "$want" ]; then
    rm -f $f.o $f.stamp
    ( $HIPCC $FLAGS $xflag -c $src -o $f.o && echo "$want" > $f.stamp ) &
    PIDS="$PIDS $!"
    BUILT="$BUILT $f"
  fi
  OBJS="$OBJS $f.o"
done
for p in $PIDS; do wait $p || { echo "build FAILED"; exit 1; }; done
if [ -n "$BUILT" ] || [ ! -f libdiqt_hip.so ]; then
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o libdiqt_hip.so $OBJS
fi
echo "built $(pwd)/libdiqt_hip.so (compiled:${BUILT:- nothing, all stamps current})"
