#!/bin/bash
# dropin_build.sh -- compile the REFERENCE's own, unmodified test programs against this repository's
# headers and libraries (authoring container only: needs /root/reference).
#
# The reference tests include "../src/socp/shooting.hpp" and "../src/models/<m>/<m>.hpp" relative to
# their own directory.  A scratch tree with two symlinks makes those paths resolve to the host mirror:
#     oracle/_ref/dropin/tests/<prog>.cpp -> /root/reference/tests/<prog>.cpp      (their source, untouched)
#     oracle/_ref/dropin/src              -> socp_amd/host/src                     (our headers)
# Outputs only under oracle/_ref/ (git-ignored).  Nothing from the reference is copied.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
REF="${REF:-/root/reference}"
OUT="$ROOT/oracle/_ref/dropin"
[ -d "$REF/tests" ] || { echo "reference tree not present: nothing to do"; exit 0; }
[ -f "$ROOT/socp_amd/_build/libsocp_host.so" ] || { echo "build the product first (__graft_entry__.build())"; exit 1; }
rm -rf "$OUT"; mkdir -p "$OUT/tests" "$OUT/bin"
ln -s "$ROOT/socp_amd/host/src" "$OUT/src"
for prog in testGoddard testDoubleIntegrator testDoubleIntegrator_WP testCovid19 testInterceptor; do
    ln -s "$REF/tests/$prog.cpp" "$OUT/tests/$prog.cpp"
    g++ -O2 -std=gnu++14 -w -I"$ROOT/include" -I"$ROOT/socp_amd/host/src/socp" -o "$OUT/bin/$prog" "$OUT/tests/$prog.cpp" \
        -L"$ROOT/socp_amd/_build" -lsocp_host -lsocp_hip -Wl,-rpath,"$ROOT/socp_amd/_build" -lpthread
    echo "built $OUT/bin/$prog"
done
# the scratch tree only existed to resolve the includes: remove the links so that nothing under oracle/_ref
# points at (or could be resolved into) reference sources when the directory travels to the GPU box
rm -rf "$OUT/tests" "$OUT/src"
