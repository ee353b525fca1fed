#!/bin/bash
cd "$(dirname "$0")/.."
for b in "$@"; do timeout -k 10 60 tools/halo_probe_$b || exit 1; done
