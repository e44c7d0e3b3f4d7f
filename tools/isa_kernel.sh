#!/bin/bash
# Body of one kernel from a hipcc -S listing: tools/isa_kernel.sh <file.s> <mangled-name-prefix>  (from its label to .Lfunc_end)
s=$(grep -n "^$2.*:" "$1" | head -1 | cut -d: -f1)
[ -z "$s" ] && { echo "no such kernel" >&2; exit 1; }
awk -v s=$s 'NR>=s' "$1" | awk '/^\.Lfunc_end/{exit} {print}'
