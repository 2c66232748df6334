"""mcts_q_amd - MI355X-native hot path of MCTS_Q (batched leaf Q evaluation + PER-DQN update).

HIP kernels live in csrc/ and are reached through the C ABI of lib/libmctsq_hip.so
(include/mctsq_hip.h); this package is the thin PyTorch-ROCm host side above it.
"""
from ._lib import lib, MctsqError  # noqa: F401
from .netdesc import NetSpec  # noqa: F401
