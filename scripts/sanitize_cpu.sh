#!/bin/bash
# AddressSanitizer + UBSan over the host-compilable solver code (no GPU; GPU sanitizers are not available on the pool):
#   1. the DEVICE solver's header (solver_dev.hpp) in its one-thread host build, through tests/test_devsolver_sim.py
#   2. the host MINPACK (minpack.cpp, incl. the threaded SIMD-lane factorisation), through tests/test_minpack.py
# Usage: bash scripts/sanitize_cpu.sh     (from the repo root; ~1 min)
set -e
cd "$(dirname "$0")/.."
T=$(mktemp -d)
SAN="-O1 -g -std=c++17 -ffp-contract=off -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer"
g++ $SAN -o $T/libsolver_sim_asan.so tests/tools/solver_sim.cpp
g++ $SAN -I include -I socp_amd/csrc -o $T/libminpack_asan.so socp_amd/csrc/minpack.cpp -lpthread
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
cat > $T/run.py <<PY
import sys, ctypes as C, subprocess
sys.path.insert(0, '$PWD'); sys.path.insert(0, '$PWD/tests')
import pytest
from socp_amd import capi
which = sys.argv[1]
if which == 'sim':
    # the fixture compiles the tool itself: hand it the sanitised object instead
    real = subprocess.check_call
    def fake(cmd, *a, **k):
        if cmd and cmd[0] == 'g++' and any('solver_sim.cpp' in c for c in cmd):
            out = cmd[cmd.index('-o') + 1]
            import shutil; shutil.copy('$T/libsolver_sim_asan.so', out); return 0
        return real(cmd, *a, **k)
    subprocess.check_call = fake
    sys.exit(pytest.main(['-x', '-q', 'tests/test_devsolver_sim.py', '-p', 'no:cacheprovider']))
capi.LIB_PATH = '$T/libminpack_asan.so'
class Dummy:
    argtypes = None; restype = None
    def __call__(self, *a): return 0
class Shim(C.CDLL):               # the sanitised object holds the MINPACK entry points only
    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            d = Dummy(); self.__dict__[name] = d; return d
capi.C.CDLL = Shim
sys.exit(pytest.main(['-x', '-q', 'tests/test_minpack.py', '-p', 'no:cacheprovider']))
PY
python3 $T/run.py sim
python3 $T/run.py minpack
rm -rf $T
echo "sanitizers: clean"
