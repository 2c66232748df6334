"""Drop-in module path of the reference's src/mctsq_config_env.py (EnvConfiguration, :15-73)."""
from mcts_q_amd.envconfig import EnvConfiguration  # noqa: F401
