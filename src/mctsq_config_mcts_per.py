"""Drop-in module path of the reference's src/mctsq_config_mcts_per.py (MCTSQ, MCTSQConfiguration)."""
from mcts_q_amd.mcts import MCTSQ, MCTSQConfiguration  # noqa: F401
