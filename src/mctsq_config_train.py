"""Drop-in module path of the reference's src/mctsq_config_train.py (TrainConfiguration, :14-59)."""
from mcts_q_amd.ptgdata import TrainConfiguration  # noqa: F401
