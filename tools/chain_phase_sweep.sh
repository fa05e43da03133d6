#!/bin/bash
# config 4 in scan mode with k_band_chain's phase-group experiment (band_scan_debug: groups << 4 | delay(100 ns) << 8)
for v in 0 $((0x0530)) $((0x0830)) $((0x0B30)) $((0x0E30)) $((0x0820)) $((0x0C20)) $((0x1020)) $((0x0640)) $((0x0940)); do
  echo "band_scan_debug=$v"
  TD_OPTS=band_mode=1,band_scan_debug=$v python tools/time_configs.py c4 | cut -c1-200
done
