#!/bin/bash
for v in "$@"; do
  ( IFS=','; for kv in $v; do export "$kv"; done
    for b in 4 8 16 32; do
      timeout -k 10 120 python3 bench.py --batch $b --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > /tmp/b.json 2>/tmp/b.err || { tail -3 /tmp/b.err; exit 1; }
      python3 -c "import json;d=json.load(open('/tmp/b.json'));print('$v batch',$b,'ms/step',d['ms_per_step'],'ms/img',round(d['ms_per_step']/$b,4),'Mpix/s',d['value'])"
    done )
done
