#!/bin/bash
# Product library from a SNAPSHOT of the sources: the ~4-minute hipcc run reads porefv.hip and its includes twice (device
# pass, then host pass), so editing csrc/ while it runs can pair a device image with host stubs of other sources.  The
# snapshot is taken first; the .so is moved into place only on success.  Log: /tmp/pfv_hipcc.log (last line BUILD_OK / BUILD_FAILED).
set -u
R="$(cd "$(dirname "$0")/.." && pwd)"
D="$(mktemp -d /tmp/pfv_build.XXXXXX)"
mkdir -p "$D/porepy_amd/csrc" "$D/include"
cp "$R"/porepy_amd/csrc/*.hip "$R"/porepy_amd/csrc/*.inc "$R"/porepy_amd/csrc/*.h "$D/porepy_amd/csrc/"
cp "$R"/include/*.h "$D/include/"
cd "$D/porepy_amd/csrc"
if hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC porefv.hip -o libporefv_hip.so > /tmp/pfv_hipcc.log 2>&1; then
  mv libporefv_hip.so "$R/porepy_amd/csrc/libporefv_hip.so"
  echo BUILD_OK >> /tmp/pfv_hipcc.log
else
  grep -v "argument unused" /tmp/pfv_hipcc.log | grep -m 30 "error" >> /tmp/pfv_hipcc.err
  echo BUILD_FAILED >> /tmp/pfv_hipcc.log
fi
rm -rf "$D"
