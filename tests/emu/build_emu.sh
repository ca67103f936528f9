#!/bin/bash
# TEST INFRASTRUCTURE ONLY: compile the unmodified kernel sources against the CPU lane emulator.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
CXX=${SVB_EMU_CXX:-/opt/rocm/lib/llvm/bin/clang++}
[ -x "$CXX" ] || CXX=clang++
OUT="$HERE/libsvb_emu.so"
SRCS=$(ls "$ROOT"/neuralsvb_amd/csrc/*.hip)
OBJS=""
mkdir -p "$HERE/build"
for s in $SRCS; do
  o="$HERE/build/$(basename "$s" .hip).o"
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ -n "$(find "$ROOT/neuralsvb_amd/csrc" "$ROOT/include" "$HERE/include" -name '*.h' -newer "$o")" ]; then
    "$CXX" -x c++ -std=c++17 -O2 -fPIC -I"$HERE/include" -I"$ROOT/include" -I"$ROOT/neuralsvb_amd/csrc" -Wno-unknown-attributes -c "$s" -o "$o" &
  fi
  OBJS="$OBJS $o"
done
wait
"$CXX" -std=c++17 -O2 -fPIC -I"$HERE/include" -c "$HERE/emu_runtime.cpp" -o "$HERE/build/emu_runtime.o"
"$CXX" -shared -o "$OUT" $OBJS "$HERE/build/emu_runtime.o"
echo "$OUT"
