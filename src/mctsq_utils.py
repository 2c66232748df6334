"""Drop-in module path of the reference's src/mctsq_utils.py: data loading, preprocessing, the run banner
(:25-541), and the entry scripts' environment factory / validation callback / result plot (:542-642; gymnasium and
matplotlib are imported when those are called)."""
from mcts_q_amd.ptgdata import (CallbackVal, Preprocessing, config_print, create_envs, import_data,  # noqa: F401
                                import_market_data, initial_print, load_data, plot_results)
