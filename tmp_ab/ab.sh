#!/bin/bash
# usage: ab.sh name1 name2 ... ; runs bench for each tmp_ab/<name>.so on the same box
for n in "$@"; do
  cp tmp_ab/$n.so fastsk_amd/lib/libfastsk_amd.so
  python bench.py --no-cpu-baseline --no-also --steps 2 --warmup 1 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$n', 'value %.2f' % d['value'], d['phases_ms_per_step'])
"
done
