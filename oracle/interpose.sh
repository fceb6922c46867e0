#!/bin/sh
# TEST INFRASTRUCTURE. Makes functions of a compiled reference object replaceable without touching its source or its machine code:
#     interpose.sh in.o out.o  MANGLED=ALIAS [MANGLED=ALIAS ...]
# For every pair the global function symbol MANGLED of in.o becomes WEAK (so that a strong definition of the same name in another
# translation unit - oracle/ref_ts_gpu_harness.cc - wins at link time, for callers in every object including in.o itself: the reference is
# compiled -fPIC with gcc, whose calls to default-visibility functions go through the symbol) and a second global symbol ALIAS is added at
# the same section offset, so that the replaced code stays callable under that name (the harness runs it as the fall-back and as the
# value its GPU-backed replacement is compared against). Nothing is stubbed, renamed away or deleted.
set -e
in="$1"; out="$2"; shift 2
args=""
for pair in "$@"; do
    sym="${pair%%=*}"; alias="${pair#*=}"
    line=$(objdump -t "$in" | awk -v s="$sym" '$NF == s && $2 == "g" && $3 == "F" { print $1, $4 }')
    [ -n "$line" ] || { echo "interpose.sh: $sym is not a global function of $in" >&2; exit 1; }
    value="${line%% *}"; section="${line#* }"
    args="$args --weaken-symbol=$sym --add-symbol $alias=$section:0x$value,global,function"
done
# shellcheck disable=SC2086
objcopy $args "$in" "$out"
