#!/bin/bash
# A/B build: tools/build_variant.sh <name> <extra hipcc flags for rnn.hip and decoder.hip>  ->  tools/ab/<name>.so
# (run with SSASR_LIB=$PWD/tools/ab/<name>.so; python -m ss_asr_amd.build must have run: the other objects are copied)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p /tmp/ssasr_var_$name
objs=""
for f in ss_asr_amd/csrc/*.hip; do
  o=/tmp/ssasr_var_$name/$(basename $f .hip).o
  b=$(basename $f)
  if [ "$b" = "rnn.hip" ] || [ "$b" = "decoder.hip" ]; then
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wno-pass-failed "$@" -I ss_asr_amd/csrc -c $f -o $o &
  else
    cp ss_asr_amd/csrc/$(basename $f .hip).o $o
  fi
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/$name.so $objs
echo tools/ab/$name.so
