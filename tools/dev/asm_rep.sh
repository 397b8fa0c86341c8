#!/bin/bash
# usage: asm_rep.sh <n> [env assignments...]
n=$1; shift
for i in $(seq 1 $n); do env "$@" ASMPW_SCALE_NOREF=1 timeout 300 python tools/dev/asmpw_scale.py 20000 8000 5000000 2 mecat2asmpw 64 1 2>&1 | tail -1 | python -c "
import sys,json,re
d=json.loads(sys.stdin.read()); t=d['device_times'][0]
m=re.search(r'candidates ([0-9.]+)',t); print('%.2f wall, candidates %s' % (d['device_seconds'], m.group(1)))"; done
