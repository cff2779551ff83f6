#!/bin/bash
# Developer A/B builds: tools/build_variant.sh <name> "<extra hipcc flags>" [git-rev]
# -> build/<name>/libmrgs.so from the working tree's (or <git-rev>'s) csrc, compiled with the extra flags.  Select it at run time with
#    MRGS_LIB=build/<name>/libmrgs.so (materialrefgs_amd/_lib.py; never a fallback).  build/ is git-ignored and travels with gpurun.
set -e
N=$1; X=$2; REV=$3
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/build/${N}_src
rm -rf $S; mkdir -p $S $R/build/$N
if [ -n "$REV" ]; then (cd $R && git archive $REV materialrefgs_amd/csrc include) | tar -x -C $S
else mkdir -p $S/materialrefgs_amd $S/include; cp -r $R/materialrefgs_amd/csrc $S/materialrefgs_amd/; cp $R/include/*.h $S/include/; rm -f $S/materialrefgs_amd/csrc/*.o $S/materialrefgs_amd/csrc/*.so; fi
make -C $S/materialrefgs_amd/csrc -j8 EXTRA="$X" > $S/build.log 2>&1 || { tail -20 $S/build.log; exit 1; }
cp $S/materialrefgs_amd/csrc/libmrgs.so $R/build/$N/libmrgs.so
echo "built build/$N/libmrgs.so ($X)"
