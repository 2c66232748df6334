"""Drop-in module path of the reference's src/mctsq_config_dqn.py (non-_per flavour: sin/cos time
features appended inside the kernels, hard target copy), served by mcts_q_amd.dqn / mcts_q_amd.encoders."""
from mcts_q_amd.dqn import DQNModelStd as DQNModel  # noqa: F401
from mcts_q_amd.dqn import PrioritizedReplayBuffer, get_activation  # noqa: F401
from mcts_q_amd.dqn import TripleEncoderDQNStd as TripleEncoderDQN  # noqa: F401
from mcts_q_amd.encoders import (ConvAttentionEnc, GasEUAEncoder, GRUAttentionEnc, TransformerEnc,  # noqa: F401
                                 add_time_features, add_time_features_to_gas_eua)
