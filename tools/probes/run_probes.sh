#!/bin/bash
# runs every built variant twice, interleaved (box-to-box and run-to-run noise shows in the spread)
cd "$(dirname "$0")/bin"
for rep in 1 2; do
  for b in w8_*; do timeout -k 10 60 ./$b "$@" || echo "$b failed"; done
done
