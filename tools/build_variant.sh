#!/bin/bash
# Diagnostic build of the library with extra compiler flags: tools/build_variant.sh <tag> [flags...]
# -> superresolution_amd/variants/libsrmi355_<tag>.so (use with SRMI_LIB=...)
set -e
tag=$1; shift
cd "$(dirname "$0")/../superresolution_amd/csrc"
mkdir -p ../variants
make -j8 BUILD=build_$tag OUT=../variants/libsrmi355_$tag.so EXTRA="-DSRMI_DIAG $*" 2>&1 | grep -E "error|warning: v" || true
ls -la ../variants/libsrmi355_$tag.so
