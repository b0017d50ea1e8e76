#!/bin/bash
# Build both libraries (product: gfx950 HIP; test infrastructure: host emulation of the same sources).
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R/porepy_amd/csrc"
mkdir -p "$R/oracle/_build"
g++ -x c++ -DPFV_EMULATE -O2 -std=c++17 -shared -fPIC -Wno-maybe-uninitialized porefv.hip -o "$R/oracle/_build/libporefv_emul.so"
if [ "$1" != "--emul-only" ]; then
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC porefv.hip -o libporefv_hip.so.tmp 2> /tmp/pfv_hipcc.log || { grep -v "argument unused" /tmp/pfv_hipcc.log | head -40; echo "HIPCC FAILED"; rm -f libporefv_hip.so.tmp; exit 1; }
  mv libporefv_hip.so.tmp libporefv_hip.so
fi
ls -la --time-style=full-iso libporefv_hip.so "$R/oracle/_build/libporefv_emul.so"
