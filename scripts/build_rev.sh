#!/bin/bash
# build_rev.sh <name> <git-rev|WORK> [extra -D flags...]: compile libtsamd from a revision into build/variants/<name>.so
set -e
NAME=$1; REV=$2; shift 2
ROOT=$(cd $(dirname $0)/.. && pwd)
TMP=$ROOT/build/rev_$NAME; rm -rf $TMP; mkdir -p $TMP/csrc $TMP/include $ROOT/build/variants
if [ "$REV" = "WORK" ]; then cp $ROOT/pytorch_sparse_amd/csrc/*.h* $TMP/csrc/; cp $ROOT/include/*.h $TMP/include/;
else (cd $ROOT && git archive $REV pytorch_sparse_amd/csrc include | tar -x -C $TMP && mv $TMP/pytorch_sparse_amd/csrc/* $TMP/csrc/); fi
SRCS=$(ls $TMP/csrc/*.hip)
hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -I$TMP/include -I$TMP/csrc "$@" $SRCS -o $ROOT/build/variants/$NAME.so
echo built $NAME from $REV
