#!/bin/bash
# Is the device code of two source trees the same?  (clean-ups of the kernel sources must not move an instruction.)
#   scripts/asm_diff.sh <treeA> [treeB = this repository]
# e.g. `git worktree add /tmp/before HEAD~1 && scripts/asm_diff.sh /tmp/before` -- prints per source file "identical device code" or
# the kernels that differ; exit status 1 if any does.  (scripts/kernel_asm.py does the work.)
R=$(cd "$(dirname "$0")/.." && pwd)
exec python "$R/scripts/kernel_asm.py" diff "$1" "${2:-$R}"
