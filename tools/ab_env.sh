#!/bin/bash
# A/B timing with per-variant environment: tools/ab_env.sh "<command>" "tag:ENV=V,ENV2=V" ...   (tag "main" = in-tree library)
cmd=$1; shift
for round in 1 2 3; do
  for spec in "$@"; do
    tag=${spec%%:*}; envs=${spec#*:}; [ "$envs" = "$spec" ] && envs=""
    (
      if [ "$tag" != main ]; then export SRMI_LIB=$PWD/superresolution_amd/variants/libsrmi355_$tag.so; fi
      for kv in ${envs//,/ }; do export "$kv"; done
      echo "== $spec (round $round)"
      timeout -k 10 200 $cmd 2>/dev/null
    ) || exit 1
  done
done
