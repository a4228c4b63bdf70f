#!/bin/bash
# A/B of two BUILDS of libgct2.so on the same box: alternates the library under csrc/ between the current build and
# ab_libs/<name>.so (built from another revision, same ABI) and runs scripts/ab_tuning.py 0 with each, N times.
# usage: bash scripts/ab_libs.sh <other.so> [rounds]
set -e
other=$1; rounds=${2:-3}
lib=gan-class-transfer2_amd/csrc/libgct2.so
cp $lib /tmp/libgct2_current.so
trap 'cp /tmp/libgct2_current.so '$lib EXIT
for i in $(seq $rounds); do
  cp /tmp/libgct2_current.so $lib; echo "current: $(AB_ROUNDS=3 python scripts/ab_tuning.py 0 | grep tuning)"
  cp $other $lib;                  echo "other:   $(AB_ROUNDS=3 python scripts/ab_tuning.py 0 | grep tuning)"
done
