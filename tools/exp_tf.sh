#!/bin/bash
# transformer forward timing across library variants (mcts_q_amd/lib + tools/bin/libexp*.so)
for f in mcts_q_amd/lib/libmctsq_hip.so tools/bin/libexp*.so; do
  echo -n "$f: "
  MCTSQ_HIP_LIB=$PWD/$f timeout -k 10 200 python3 tools/bench_configs.py 2>/dev/null | grep '"transformer/transformer/gru", "batch": 16384' | cut -c1-140
done
