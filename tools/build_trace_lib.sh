#!/bin/bash
# Diagnostic build of the library with per-phase timestamps in the persistent decode loop
# (tools/dectrace.py).  Output: tools/ab/trace.so
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/ab /tmp/ssasr_trace
objs=""
for f in ss_asr_amd/csrc/*.hip; do
  o=/tmp/ssasr_trace/$(basename $f .hip).o
  extra=""
  [ "$(basename $f)" = "decoder.hip" ] && extra="-DSSASR_TRACE_BUILD $TRACE_EXTRA"
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC $extra -I ss_asr_amd/csrc -c $f -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/trace.so $objs
echo tools/ab/trace.so
