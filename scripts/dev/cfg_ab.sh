#!/bin/bash
# Developer: BASELINE configs 1-4 step times under two (or more) library builds on ONE box: scripts/dev/cfg_ab.sh libA.so libB.so ...
R=${GRAFT_REPO_ROOT:-$PWD}
for v in "$@"; do
  for K in 1 2 3 4; do
    BEAR_AMD_LIB=$R/$v timeout -k 10 400 python3 $R/scripts/baseline_configs.py configs$K 2>/dev/null | python3 -c "
import json,sys
txt=sys.stdin.read(); d=json.loads(txt[txt.index('{'):])
def walk(x,p=''):
    if isinstance(x,dict):
        for k,v in x.items(): walk(v,p+'/'+k[:60])
    elif p.endswith('us_per_step') or 'eval' in p.split('/')[-1] and isinstance(x,(int,float)): print('$v'.split('/')[-1], p, round(x,2))
walk(d)" || exit 1
  done
done
