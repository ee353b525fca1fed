#!/bin/bash
# builds (if hipcc is here) and runs tools/halo_probe_<abl> for the ablation bit sets given (default: a standard list)
cd "$(dirname "$0")/.."
LIST=${@:-0 1 2 4 8 3 6 5 12 14 15}
for a in $LIST; do
  if [ ! -x tools/halo_probe_$a ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DHALO_ABL=$a -Iswin_transformer_object_detection_amd/csrc -Iinclude tools/halo_probe.hip -o tools/halo_probe_$a || exit 1
  fi
done
for a in $LIST; do timeout -k 10 60 tools/halo_probe_$a || exit 1; done
