#!/bin/bash
# usage: profiles/tools/mkig.sh NAME "-DFLAGS" -> profiles/tools/lib_NAME.so with icholt_grid.hip rebuilt
set -e
cd /root/repo/ilupp_amd/csrc
mkdir -p scratch
F="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../../include -Wall -Wno-unused-result"
/opt/rocm/bin/hipcc $F $2 -c icholt_grid.hip -o scratch/ig_$1.o
OBJS=$(ls *.o | grep -v '^icholt_grid\.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /root/repo/profiles/tools/lib_$1.so $OBJS scratch/ig_$1.o
ls -la /root/repo/profiles/tools/lib_$1.so
