"""Drop-in module path of the reference's src/mctsq_utils.py: data loading, preprocessing and the run banner
(:25-541).  The gymnasium-based helpers of that module (CallbackVal, create_envs, :543-579) and plot_results
(:581-642, matplotlib) are outside the hot-path scope; the vectorised environment is mcts_q_amd.ptgenv.PtGBatchEnv."""
from mcts_q_amd.ptgdata import (Preprocessing, config_print, import_data, import_market_data, initial_print,  # noqa: F401
                                load_data)
