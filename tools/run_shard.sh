#!/bin/bash
# tools/run_shard.sh N [N ...] : the sharded paths of BASELINE configs[3] and configs[4] at full size on N ranks (one JSON line per N).
# On an 8-GPU node: tools/run_shard.sh 1 2 4 8     (rank r uses GPU r % device count, so a 1-GPU box rehearses any N)
for n in "$@"; do
  timeout -k 10 1100 python3 tools/run_shard.py $n ${SHARD_ARGS:-} || exit 1
done
