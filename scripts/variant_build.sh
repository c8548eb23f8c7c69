#!/bin/bash
# A variant build of the product library in ITS OWN directory:   LIB=$(bash scripts/variant_build.sh <name> [make variables ...])
#   e.g.  LIB=$(bash scripts/variant_build.sh prof_solver SOLVER_DEFS=-DSOCP_SOLVER_PROFILE)  ->  socp_amd/_build_prof_solver/libsocp_hip.so
# Prints the library's path (select it with SOCP_LIB_PATH); exit status 1 and the build log's tail on stderr when the build fails.
# The product library in socp_amd/_build is never rebuilt in place by a measurement script (a run killed at a time limit
# would leave a profiling build behind for every later test and bench step).  Objects whose flags a variant cannot change are copied
# from the product build when they are fresh, so a variant costs one translation unit.
cd "$(dirname "$0")/.."
NAME=$1; shift
[ -n "$NAME" ] || { echo "usage: variant_build.sh <name> [VAR=value ...]" >&2; exit 2; }
DIR=$PWD/socp_amd/_build_$NAME
mkdir -p $DIR
SKIP=""
for a in "$@"; do
  case "$a" in
    SOLVER_DEFS=*) SKIP="$SKIP kernels_solver.o" ;;
    FACTOR_DEFS=*) SKIP="$SKIP kernels_factor_fast.o" ;;
    *) SKIP="ALL" ;;                                        # (anything else -- COMMON, ARCH ...: every object is the variant's own)
  esac
done
if [ "$SKIP" != ALL ]; then
  for o in socp_amd/_build/*.o; do
    b=$(basename $o)
    case " $SKIP " in *" $b "*) continue ;; esac
    if [ ! -f $DIR/$b ] || [ $o -nt $DIR/$b ]; then cp -p $o $DIR/$b; fi
  done
fi
# the variant's own objects are rebuilt every time (their defines are not part of make's dependencies)
for b in $SKIP; do [ "$b" != ALL ] && rm -f $DIR/$b; done
if ! make -s -C socp_amd/csrc OUT=$DIR "$@" > $DIR/build.log 2>&1; then
  tail -20 $DIR/build.log >&2
  exit 1
fi
echo $DIR/libsocp_hip.so
