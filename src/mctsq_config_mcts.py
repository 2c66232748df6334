"""Drop-in module path of the reference's src/mctsq_config_mcts.py (non-`_per` MCTSQ, MCTSQConfiguration)."""
from mcts_q_amd.mcts_std import MCTSQ, MCTSQConfiguration  # noqa: F401
