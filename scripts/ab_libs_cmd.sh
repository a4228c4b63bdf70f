#!/bin/bash
# like scripts/ab_libs.sh for an arbitrary command: alternates csrc/libgct2.so between the current build and <other.so>, runs <cmd> with each
# usage: bash scripts/ab_libs_cmd.sh <other.so> <rounds> <cmd...>
set -e
other=$1; rounds=$2; shift 2
lib=gan-class-transfer2_amd/csrc/libgct2.so
cp $lib /tmp/libgct2_current.so
trap 'cp /tmp/libgct2_current.so '$lib EXIT
for i in $(seq $rounds); do
  cp /tmp/libgct2_current.so $lib; echo "== current"; "$@" 2>&1 | grep -v amdgpu.ids
  cp $other $lib;                  echo "== other";   "$@" 2>&1 | grep -v amdgpu.ids
done
