#!/bin/bash
for f in mcts_q_amd/lib/libmctsq_hip.so tools/bin/libexp_c*.so; do
  echo -n "$f: "
  MCTSQ_HIP_LIB=$PWD/$f timeout -k 10 200 python3 tools/bench_configs.py 2>/dev/null | grep '"conv/conv/mlp"' | cut -c1-110
done
