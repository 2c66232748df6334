#!/bin/bash
# times the fused forward with the product library and any variant libraries built into tools/bin/libexp*.so
# (make -C mcts_q_amd/csrc OUT=../../tools/bin/libexp_x.so OBJDIR=../../build/obj_x EXTRA=-D...)
for f in mcts_q_amd/lib/libmctsq_hip.so $(ls tools/bin/libexp*.so 2>/dev/null); do
  echo -n "$f: "
  MCTSQ_HIP_LIB=$PWD/$f timeout -k 10 100 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms']*1e3, 'us')"
done
