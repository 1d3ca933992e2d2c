#!/bin/bash
# Stand-in for the ROCm compiler in the sanitizer tests of the code cache: writes something that passes for a code object to the -o path.
out=""
while [ $# -gt 0 ]; do if [ "$1" = "-o" ]; then out="$2"; shift; fi; shift; done
sleep 0.05
printf '\177ELF' > "$out"
head -c 4096 /dev/zero >> "$out"
