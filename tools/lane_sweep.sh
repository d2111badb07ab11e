#!/bin/bash
# usage: tools/lane_sweep.sh VAR v1 v2 ...   (runs bench.py with VAR=value, prints value / ms)
var=$1; shift
for v in "$@"; do
  env $var=$v timeout -k 10 200 python bench.py --steps 100 --cpu-seconds 0 2>/dev/null > /tmp/ls.json || exit 1
  python - "$var=$v" <<'PY'
import json, sys
d = json.loads(open('/tmp/ls.json').read().strip().splitlines()[-1])
iso = d['roofline'].get('isolated') or {}
print(sys.argv[1], round(d['value'] / 1e6, 1), 'Mgates/s', round(d['ms_per_step'], 4), 'ms/step; isolated psd', iso.get('avg_stage_ms'), 'total', iso.get('device_total_ms'))
PY
done
