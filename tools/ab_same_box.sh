#!/bin/bash
# A/B of engine builds on ONE GPU box (box-to-box variance is ~2 %): put candidate libraries in
# tmp_ab/<name>.so, then `gpurun -- tools/ab_same_box.sh base cand base cand`. Prints the headline
# bench's combos/s and per-phase milliseconds for each, in the order given.
for n in "$@"; do
  cp tmp_ab/$n.so fastsk_amd/lib/libfastsk_amd.so
  python bench.py --no-cpu-baseline --no-also --steps 2 --warmup 1 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$n', 'value %.2f' % d['value'], d['phases_ms_per_step'])
"
done
