#!/bin/bash
for n in "$@"; do
  cp tmp_ab/$n.so fastsk_amd/lib/libfastsk_amd.so
  echo "== $n"; python tools/bench_configs.py 2>/dev/null | grep cfg1 | cut -c1-200
done
