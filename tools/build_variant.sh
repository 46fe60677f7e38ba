#!/bin/bash
# build container: one more build of the library into build/<name>.so with extra hipcc flags (A/B and stage-timing builds)
# usage: tools/build_variant.sh <name> [flags...]
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -ffp-contract=on "$@" \
  -o build/$name.so baryonforge_amd/csrc/bfg_mi355.hip 2>&1 | grep -E "error|Error" ; ls -la build/$name.so
