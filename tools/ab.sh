#!/bin/bash
# A/B timing of library variants on ONE device, alternating processes: tools/ab.sh "<command>" tagA tagB ...
# (tag "main" = the in-tree library)
cmd=$1; shift
for round in 1 2 3; do
  for tag in "$@"; do
    if [ "$tag" = main ]; then unset SRMI_LIB; else export SRMI_LIB=$PWD/superresolution_amd/variants/libsrmi355_$tag.so; fi
    echo "== $tag (round $round)"
    timeout -k 10 200 $cmd 2>/dev/null || exit 1
  done
done
