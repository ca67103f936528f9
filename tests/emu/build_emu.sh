#!/bin/bash
# TEST INFRASTRUCTURE ONLY: compile the unmodified kernel sources against the CPU lane emulator.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
CXX=${SVB_EMU_CXX:-/opt/rocm/lib/llvm/bin/clang++}
[ -x "$CXX" ] || CXX=clang++
OUT="$HERE/libsvb_emu.so"
mkdir -p "$HERE/build"
exec 9> "$HERE/build/.lock"          # pytest-xdist workers build one at a time; the library appears by an atomic rename
flock 9
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
RT="$HERE/build/emu_runtime.o"
if [ ! -f "$RT" ] || [ "$HERE/emu_runtime.cpp" -nt "$RT" ] || [ -n "$(find "$HERE/include" -name '*.h' -newer "$RT")" ]; then
  "$CXX" -std=c++17 -O2 -fPIC -I"$HERE/include" -c "$HERE/emu_runtime.cpp" -o "$RT"
fi
NEWER=""
for o in $OBJS "$RT"; do [ -f "$OUT" ] && [ ! "$o" -nt "$OUT" ] || NEWER=1; done
if [ -n "$NEWER" ]; then
  "$CXX" -shared -o "$OUT.tmp.$$" $OBJS "$RT"
  mv -f "$OUT.tmp.$$" "$OUT"
fi
echo "$OUT"
