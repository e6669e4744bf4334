#!/bin/bash
# usage: tools/build_gen_variant.sh <name> [GEN_T_LEAD=.. GEN_T_RING=.. GEN_MIX=..]   -- a VARIANT of the library whose term-loop rounds come from
# tools/gen/gen_pair_round.py with other knobs (table lead / ring, every n-th coefficient pair as a DPP operand) -> gpurun_variants/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
W=/tmp/genvar_$NAME; rm -rf $W; mkdir -p $W gpurun_variants
cp jpegsnoop_amd/csrc/*.h $W/
env "$@" python tools/gen/gen_pair_round.py $W/jsnoop_pair_round.h
cp jpegsnoop_amd/csrc/jsnoop_kernels.hip $W/k.hip
if echo "$@" | grep -q GEN_MIX; then python - $W/k.hip <<'PY'
import sys
p=sys.argv[1]; s=open(p).read()
s=s.replace('[ad] "=&v"(ad), [rw] "=&v"(rw)','[ad] "=&v"(ad), [rw] "=&v"(rw), [ey] "=&v"(ey)')
s=s.replace('[ah] "v"(L.a_half), [arw] "v"(L.a_rw)','[ah] "v"(L.a_half), [aey] "v"(L.a_rw - 256u), [arw] "v"(L.a_rw)')
s=s.replace('uint32_t ad, rw;','uint32_t ad, rw; float ey;')
open(p,'w').write(s)
PY
fi
make -s -j8 -C jpegsnoop_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wno-unused-function -Wno-unused-value -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS -x hip -c $W/k.hip -I$W -Iinclude -o $W/k.o
OBJS=$(ls jpegsnoop_amd/csrc/build/*.o | grep -v jsnoop_kernels.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_variants/lib_$NAME.so $W/k.o $OBJS -L/opt/rocm/lib -lrocprofiler-sdk-roctx -Wl,-rpath,/opt/rocm/lib
echo built gpurun_variants/lib_$NAME.so
