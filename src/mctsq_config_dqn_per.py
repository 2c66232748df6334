"""Drop-in module path of the reference's src/mctsq_config_dqn_per.py: the same public names, served by
the MI355X HIP path (mcts_q_amd.dqn, mcts_q_amd.encoders).  `from src.mctsq_config_dqn_per import DQNModel`
(src/mctsq_config_mcts_per.py:29 in the reference) resolves here when this repository is the cwd."""
from mcts_q_amd.dqn import (DQNModel, PrioritizedReplayBuffer, TripleEncoderDQN, get_activation)  # noqa: F401
from mcts_q_amd.encoders import (ConvAttentionEnc, GasEUAEncoder, GRUAttentionEnc, TransformerEnc)  # noqa: F401
