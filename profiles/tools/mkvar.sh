#!/bin/bash
# usage: profiles/tools/mkvar.sh NAME FILE.hip "-DFLAGS"  -> profiles/tools/lib_NAME.so : FILE.hip rebuilt with the flags, the other objects as they are
# (select it with ILUPP_HIP_LIBRARY=profiles/tools/lib_NAME.so; delete it afterwards: every gpurun call pushes it)
set -e
cd /root/repo/ilupp_amd/csrc
mkdir -p scratch
F="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../../include -Wall -Wno-unused-result"
B=$(basename $2 .hip)
/opt/rocm/bin/hipcc $F $3 -c $2 -o scratch/${B}_$1.o
OBJS=$(ls *.o | grep -v "^${B}\.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /root/repo/profiles/tools/lib_$1.so $OBJS scratch/${B}_$1.o
ls -la /root/repo/profiles/tools/lib_$1.so
