"""Drop-in module path of the reference's src/mctsq_opt.py (calculate_optimum, mctsq_opt.py:33-39)."""
from mcts_q_amd.opt import calculate_optimum  # noqa: F401
