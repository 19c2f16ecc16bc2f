#!/bin/bash
# measurement builds of the blocks kernels (FSK_SXB_DBG) under rocprofv3: per-kernel times
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out/r06"; cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  O=$R/gpurun_out/r06/dbg_$v
  LIBARG=""; [ "$v" != "base" ] && LIBARG="--lib $R/fastsk_amd/lib/lib$v.so"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -- python3 "$R/tools/bench_sparse_large_n.py" --only protein_like_64k --combos 20 $LIBARG > "$O.out" 2> "$O.err"
  echo "== $v"; python3 "$R/tools/kstats.py" "$O" | head -5
done
