#!/bin/bash
# usage: profiles/tools/mkwx.sh NAME "-DFLAGS"  -> profiles/tools/lib_NAME.so : st_wave.hip rebuilt with the flags, the other objects as they are
set -e
cd /root/repo/ilupp_amd/csrc
mkdir -p scratch
F="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../../include -Wall -Wno-unused-result"
/opt/rocm/bin/hipcc $F $2 -c st_wave.hip -o scratch/stw_$1.o
OBJS=$(ls *.o | grep -v '^st_wave\.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /root/repo/profiles/tools/lib_$1.so $OBJS scratch/stw_$1.o
ls -la /root/repo/profiles/tools/lib_$1.so
