"""Training configuration, data loading and preprocessing: host-side mirror of TrainConfiguration
(/root/reference/src/mctsq_config_train.py:14-59) and of import_market_data / import_data / load_data / Preprocessing
(/root/reference/src/mctsq_utils.py:25-493) - SURVEY.md 8f-4.  Start-up plumbing (pandas / NumPy, runs once), kept
so the reference's entry scripts find the same names; `Preprocessing.dict_env_kwargs` is also what builds the
vectorised environment (mcts_q_amd/ptgenv.py).  Same attribute names, dictionary keys, array shapes and error
behaviour; YAML and CSV files are the USER'S (paths as in the reference), nothing of them ships here.

Differences: the look-ahead arrays are built with one sliding-window view instead of a loop per offset; the
configuration classes accept `values=` dicts besides the YAML paths, and load_data accepts a `reader` for tests.
"""
from __future__ import annotations

import warnings
from typing import Any

import numpy as np
import yaml

from .envconfig import EnvConfiguration
from .opt import calculate_optimum

MARKET_COLUMNS = {"el": ("Day-Ahead-price [Euro/MWh]", 0.1), "gas": ("THE_DA_Gas [Euro/MWh]", 0.1),
                  "eua": ("EUA_CO2 [Euro/t]", 1.0)}
PROCESS_COLUMNS = ("Time [s]", "T_cat [gradC]", "n_h2 [mol/s]", "n_ch4 [mol/s]", "n_h2_res [mol/s]", "m_DE [kg/h]",
                   "Pel [W]")
OP_DATA_FILES = (("startup_cold", 2), ("startup_hot", 3), ("cooldown", 4), ("standby_down", 5), ("standby_up", 6),
                 ("op1_start_p", 7), ("op2_start_f", 8), ("op3_p_f", 9), ("op4_p_f_p_5", 10), ("op5_p_f_p_10", 11),
                 ("op6_p_f_p_15", 12), ("op7_p_f_p_22", 13), ("op8_f_p", 14), ("op9_f_p_f_5", 15),
                 ("op10_f_p_f_10", 16), ("op11_f_p_f_15", 17), ("op12_f_p_f_20", 18))
MIN_TRAIN_LEN = 6           # days kept free at the end of the training set for the day-ahead window (:158-161)


class TrainConfiguration:
    """config/config_train.yaml -> attributes; seq_length / seq_step come from config/config_mctsq.yaml and must
    cover one simulation step (mctsq_config_train.py:44-59)."""

    def __init__(self, path: str = "config/config_train.yaml", mctsq_path: str = "config/config_mctsq.yaml",
                 env_path: str = "config/config_env.yaml", values: dict | None = None, mctsq_values: dict | None = None,
                 env_values: dict | None = None):
        def load(p, v):
            if v is not None:
                return v
            with open(p, "r", encoding="utf-8") as f:
                return yaml.safe_load(f)
        values = load(path, values)
        self.__dict__.update(values)
        self.path, self.slurm_id = None, None
        com_set = ['pc', 'slurm']
        if self.com_conf not in com_set:
            raise ValueError("Invalid computation setup specified - data/config_train.yaml -> com_conf "
                             f": {self.com_conf} must match {com_set}")
        assert len(self.r_seed_train) == len(self.r_seed_test), \
            'Training and test sets must have the same number of random seeds!'
        self.seed_train, self.seed_test = None, None
        mc = load(mctsq_path, mctsq_values)
        self.seq_length, self.seq_step = mc["seq_length"], mc["seq_step"]
        ec = load(env_path, env_values)
        seq_total = self.seq_length * self.seq_step * ec["time_step_op"]
        assert seq_total >= ec["sim_step"], (f'Sequence {seq_total} must be equal or smaller than one simulation step '
                                             f'{ec["sim_step"]} in the environment!')


def _read_csv(file_path: str):
    import pandas as pd
    return pd.read_csv(file_path, delimiter=";", decimal=".")


def import_market_data(csvfile: str, market_type: str, path: str, reader=_read_csv) -> np.ndarray:
    """Day-ahead prices of one market as a float array: electricity and gas in ct/kWh (Euro/MWh / 10), EUA in
    Euro/t (mctsq_utils.py:25-52)."""
    if market_type not in MARKET_COLUMNS:
        raise ValueError(f"Invalid market data type {market_type}. Must be one of ['el', 'gas', 'eua']!")
    df = reader(path + "/" + csvfile)
    import pandas as pd
    pd.to_datetime(df["Time"], format="%d-%m-%Y %H:%M")            # the reference parses (and so validates) the stamps
    column, _ = MARKET_COLUMNS[market_type]
    arr = df[column].values.astype(float)
    return arr / 10 if market_type != "eua" else arr


def import_data(csvfile: str, path: str, reader=_read_csv) -> np.ndarray:
    """One methanation data set as [rows, 7]: time, T_cat, n_h2, n_ch4, n_h2_res, m_DE, Pel (mctsq_utils.py:54-82)."""
    df = reader(path + "/" + csvfile)
    return np.c_[tuple(df[c].values.astype(float) for c in PROCESS_COLUMNS)]


def check_training_length(env_config, n_train_days: int) -> None:
    """train_len_d and its divisibility by the episode length (mctsq_utils.py:156-179)."""
    env_config.train_len_d = n_train_days - MIN_TRAIN_LEN
    assert env_config.train_len_d > 0, f'The training set size must be greater than {MIN_TRAIN_LEN} days'
    if env_config.train_len_d % env_config.eps_len_d != 0:
        divisors = [i for i in range(1, env_config.train_len_d + 1) if env_config.train_len_d % i == 0]
        assert False, (f'The training set size {env_config.train_len_d} must be divisible by the episode length - '
                       f'data/config_env.yaml -> eps_len_d : {env_config.eps_len_d}; Possible divisors are: {divisors}')
    assert env_config.train_len_d >= env_config.eps_len_d, \
        (f'Training set size ({env_config.train_len_d}) must be larger or at least equal to the defined episode '
         f'length ({env_config.eps_len_d})!')


def apply_scenario(env_config, dict_price_data: dict) -> None:
    """Scenario 2: fixed SNG price; scenario 3: no gas and no EUA revenue (mctsq_utils.py:141-149)."""
    if env_config.scenario in [2, 3]:
        gas_price = env_config.ch4_price_fix if env_config.scenario == 2 else 0
        for key in ('gas_price_train', 'gas_price_val', 'gas_price_test'):
            dict_price_data[key] = np.full(len(dict_price_data[key]), gas_price)
        if env_config.scenario == 3:
            for key in ('eua_price_train', 'eua_price_val', 'eua_price_test'):
                dict_price_data[key] = np.zeros(len(dict_price_data[key]))


def load_data(env_config, train_config, reader=_read_csv):
    """-> (dict_price_data, dict_op_data) with the reference's keys (mctsq_utils.py:84-181)."""
    print("---Load data")
    path = train_config.path
    dict_price_data = {}
    for name, short in (('el_price', 'el'), ('gas_price', 'gas'), ('eua_price', 'eua')):
        for split in ('train', 'val', 'test'):
            dict_price_data[f'{name}_{split}'] = import_market_data(
                getattr(env_config, f'datafile_path_{split}_{short}'), short, path, reader)
    for split in ('train', 'val', 'test'):
        el_h = len(dict_price_data[f'el_price_{split}'])
        sizes = {el_h // 24, len(dict_price_data[f'gas_price_{split}']), len(dict_price_data[f'eua_price_{split}'])}
        if len(sizes) > 1:
            warnings.warn(f"Market data size does not match for {split}: electricity ({el_h}h = {el_h // 24}d), "
                          f"gas ({len(dict_price_data[f'gas_price_{split}'])}d), and EUA "
                          f"({len(dict_price_data[f'eua_price_{split}'])}d)! -> Check size!", UserWarning)
    dict_op_data = {key: import_data(getattr(env_config, f'datafile_path{num}'), path, reader)
                    for key, num in OP_DATA_FILES}
    apply_scenario(env_config, dict_price_data)
    dict_price_data.update({f'{key}_reward_level': env_config.r_0_values[key]
                            for key in ('el_price', 'gas_price', 'eua_price')})
    check_training_length(env_config, len(dict_price_data['gas_price_train']))
    return dict_price_data, dict_op_data


def _lookahead(series, total_len: int) -> np.ndarray:
    """out[i, t] = series[t + i] for i < total_len, t < len - total_len (the loop of mctsq_utils.py:314-323)."""
    series = np.asarray(series, dtype=np.float64)
    win = np.lib.stride_tricks.sliding_window_view(series, total_len)      # [len - total_len + 1, total_len]
    return np.ascontiguousarray(win[:len(series) - total_len].T)


class Preprocessing:
    """Potential rewards, look-ahead arrays, episode bookkeeping and the environment kwargs
    (mctsq_utils.py:183-493)."""

    _ENV_KEYS = (
        "noise", "eps_len_d", "sim_step", "time_step_op", "price_ahead", "price_past", "scenario", "convert_mol_to_Nm3",
        "H_u_CH4", "H_u_H2", "dt_water", "cp_water", "rho_water", "Molar_mass_CO2", "Molar_mass_H2O", "h_H2O_evap",
        "eeg_el_price", "heat_price", "o2_price", "water_price", "min_load_electrolyzer", "max_h2_volumeflow", "eta_CHP",
        "t_cat_standby", "t_cat_startup_cold", "t_cat_startup_hot", "time1_start_p_f", "time2_start_f_p", "time_p_f",
        "time_f_p", "time1_p_f_p", "time2_p_f_p", "time23_p_f_p", "time3_p_f_p", "time34_p_f_p", "time4_p_f_p",
        "time45_p_f_p", "time5_p_f_p", "time1_f_p_f", "time2_f_p_f", "time23_f_p_f", "time3_f_p_f", "time34_f_p_f",
        "time4_f_p_f", "time45_f_p_f", "time5_f_p_f", "i_fully_developed", "j_fully_developed", "el_l_b", "el_u_b",
        "gas_l_b", "gas_u_b", "eua_l_b", "eua_u_b", "T_l_b", "T_u_b", "h2_l_b", "h2_u_b", "ch4_l_b", "ch4_u_b",
        "h2_res_l_b", "h2_res_u_b", "h2o_l_b", "h2o_u_b", "heat_l_b", "heat_u_b")

    def __init__(self, dict_price_data: dict, dict_op_data: dict, env_config, train_config) -> None:
        self.env_config, self.train_config = env_config, train_config
        self.dict_price_data, self.dict_op_data = dict_price_data, dict_op_data
        self.dict_pot_r_b = None
        self.r_level = None
        self.e_r_b_train = self.e_r_b_val = self.e_r_b_test = None
        self.g_e_train = self.g_e_val = self.g_e_test = None
        self.eps_sim_steps_train = self.eps_sim_steps_val = self.eps_sim_steps_test = None
        self.eps_ind = None
        self.overhead_factor = 10
        assert isinstance(self.overhead_factor, int), f"Episode overhead ({self.overhead_factor}) must be an integer!"
        self.n_eps = self.num_loops = self.n_eps_loops = None
        self.preprocessing_rew()
        self.preprocessing_array()
        self.define_episodes()

    def preprocessing_rew(self) -> None:
        """T-OPT potential reward and load identifier per hour of every split (:251-299)."""
        print("---Calculate the theoretical optimum, the potential reward, and the load identifier")
        d, names = self.dict_price_data, self.env_config.stats_names
        self.dict_pot_r_b = {}
        for split, label in (("train", "Training"), ("val", "Validation"), ("test", "Test")):
            s = calculate_optimum(d[f'el_price_{split}'], d[f'gas_price_{split}'], d[f'eua_price_{split}'], label, names,
                                  env_config=self.env_config)
            self.dict_pot_r_b[f'pot_rew_{split}'] = s['Meth_reward_stats']
            self.dict_pot_r_b[f'part_full_b_{split}'] = s['part_full_stats']
        level = calculate_optimum(d['el_price_reward_level'], d['gas_price_reward_level'], d['eua_price_reward_level'],
                                  "reward_Level", names, env_config=self.env_config)
        self.r_level = level['Meth_reward_stats']

    def preprocessing_array(self) -> None:
        """e_r_b_* [3, price_ahead + price_past, hours - window]: price, potential reward, load flag seen `i` hours
        after entry t; g_e_* [2, 2, days - 1]: gas / EUA today and tomorrow (:302-345)."""
        total_len = self.env_config.price_ahead + self.env_config.price_past
        d, r = self.dict_price_data, self.dict_pot_r_b
        for split in ("train", "val", "test"):
            e = np.stack([_lookahead(d[f'el_price_{split}'], total_len), _lookahead(r[f'pot_rew_{split}'], total_len),
                          _lookahead(r[f'part_full_b_{split}'], total_len)])
            gas, eua = np.asarray(d[f'gas_price_{split}'], dtype=np.float64), np.asarray(d[f'eua_price_{split}'], dtype=np.float64)
            g = np.stack([np.stack([gas[:-1], gas[1:]]), np.stack([eua[:-1], eua[1:]])])
            setattr(self, f'e_r_b_{split}', e)
            setattr(self, f'g_e_{split}', g)

    def define_episodes(self) -> None:
        """Episode counts and steps per split (:347-381)."""
        print("---Define episodes and step size limits")
        ec = self.env_config
        val_len_d = len(self.dict_price_data['gas_price_val']) - 1
        test_len_d = len(self.dict_price_data['gas_price_test']) - 1
        self.n_eps = int(ec.train_len_d / ec.eps_len_d)
        self.eps_len = 24 * 3600 * ec.eps_len_d
        self.eps_sim_steps_train = int(self.eps_len / ec.sim_step)
        self.eps_sim_steps_val = int(24 * 3600 * val_len_d / ec.sim_step)
        self.eps_sim_steps_test = int(24 * 3600 * test_len_d / ec.sim_step)
        self.num_loops = self.train_config.train_steps / (self.eps_sim_steps_train * self.n_eps)
        print("    > Total number of training steps =", self.train_config.train_steps)
        print("    > No. of loops over the entire training set =", round(self.num_loops, 3))
        print("    > Training steps per episode =", self.eps_sim_steps_train)
        print("    > Steps in the evaluation set =", self.eps_sim_steps_test, "\n")
        self.rand_eps_ind()
        self.n_eps_loops = self.n_eps * int(self.num_loops)

    def rand_eps_ind(self) -> None:
        """Episode order: every loop over the training set is an independent shuffle of its n_eps subsets, drawn from
        the global numpy generator seeded with seed_train (:383-410)."""
        np.random.seed(self.train_config.seed_train)
        ec = self.env_config
        if ec.train_len_d == ec.eps_len_d:
            self.eps_ind = np.zeros(self.n_eps * int(self.num_loops) * self.overhead_factor)
            return
        loops = (1 if self.num_loops < 1 else int(self.num_loops)) * self.overhead_factor
        order = np.tile(np.linspace(start=0, stop=self.n_eps - 1, num=self.n_eps), (loops, 1))
        for row in order:
            np.random.shuffle(row)
        self.eps_ind = order.reshape(self.n_eps * loops).astype(int)

    def dict_env_kwargs(self, train_val_test_type: str = "train") -> dict[str, Any]:
        """The `dict_input` of the environment for one split (:412-493)."""
        ec = self.env_config
        kw = {f"ptg_{key}": ec.ptg_state_space[key] for key in ("standby", "cooldown", "startup", "partial_load", "full_load")}
        kw.update({key: getattr(ec, key) for key in self._ENV_KEYS})
        kw.update(n_eps_loops=self.n_eps_loops, reward_level=self.r_level, action_type="discrete")
        kw.update(self.dict_op_data)
        kw.update(seq_length=self.train_config.seq_length, seq_step=self.train_config.seq_step,
                  rew_l_b=np.min(self.e_r_b_train[1, 0, :]), rew_u_b=np.max(self.e_r_b_train[1, 0, :]))
        if train_val_test_type == "train":
            kw.update(eps_ind=self.eps_ind, state_change_penalty=ec.state_change_penalty,
                      eps_sim_steps=self.eps_sim_steps_train, e_r_b=self.e_r_b_train, g_e=self.g_e_train)
        elif train_val_test_type in ("val", "test"):
            kw.update(eps_ind=None, state_change_penalty=0.0,
                      eps_sim_steps=getattr(self, f"eps_sim_steps_{train_val_test_type}"),
                      e_r_b=getattr(self, f"e_r_b_{train_val_test_type}"), g_e=getattr(self, f"g_e_{train_val_test_type}"))
        else:
            raise ValueError(f'Invalid type: {train_val_test_type}. Must be "train", "val", or "test".')
        return kw


def initial_print() -> None:
    """The banner of the reference's entry scripts (mctsq_utils.py:495-499)."""
    print('\n---------------------------------------MCTS_Q----------------------------------------')
    print('--Monte Carlo Tree Search and Deep Q Network for Power-to-Gas dispatch optimization--')
    print('-------------------------------------------------------------------------------------\n')


def config_print(env_config, train_config, mcts_config) -> str:
    """Run identifier MCTS_Q_<inv>_BS<scenario>_<OP>_ep<days>_ts<step><hyper>_rs<seed> (mctsq_utils.py:501-541)."""
    cases = {1: "1 - Trading at the electricity, gas, and emission spot markets",
             2: "2 - Fixed synthetic natural gas (SNG) price and trading at the electricity and emission spot markets",
             3: "3 - Participating in EEG tenders by using a CHP plant and trading at the electricity spot markets"}
    print("Set training case...")
    print(f"---Training case details: MCTS_Q_{train_config.str_inv} ")
    print(f"    > Business case (_BS):\t\t\t {cases[3 if env_config.scenario not in (1, 2) else env_config.scenario]}")
    print(f"    > Operational load level (_OP) :\t\t {env_config.operation}")
    print(f"    > Training episode length (_ep) :\t\t {env_config.eps_len_d} days")
    print(f"    > Time step size (action frequency) (_ts) :\t {env_config.sim_step} seconds")
    print(f"    > Random seed (_rs) :\t\t\t {train_config.seed_train}")
    str_id = (f"MCTS_Q_{train_config.str_inv}_BS{env_config.scenario}_{env_config.operation}_ep{env_config.eps_len_d}"
              f"_ts{env_config.sim_step}")
    return str_id + mcts_config.get_hyper() + "_rs" + str(train_config.seed_train)


# ---- gymnasium / matplotlib helpers of the entry scripts (mctsq_utils.py:542-642) -------------------------------
# Both libraries are imported inside the functions, so the module (and `from src.mctsq_utils import ...` at the top
# of mctsq_main*.py) imports where they are absent; calling them without the library raises its ImportError.
class CallbackVal:
    """Validation hook of `MCTSQ.learn` (mctsq_utils.py:542-549): the validation environment, the interval in
    training steps and the statistics lists the agent fills."""

    def __init__(self, env_id, env_kwargs_data_val, val_steps):
        import gymnasium as gym
        self.env = gym.make(env_id, dict_input=env_kwargs_data_val, train_or_eval="eval")
        self.val_steps = val_steps
        self.stats = {'steps': [], 'cum_rew': []}
        self.model = None


def create_envs(env_id: str, env_kwargs_data: dict, train_config):
    """(training environment, validation callback, test environment) (mctsq_utils.py:551-575)."""
    import gymnasium as gym
    make = lambda split, mode: gym.make(env_id, dict_input=env_kwargs_data[f'env_kwargs_{split}'], train_or_eval=mode)
    return (make('train', "train"),
            CallbackVal(env_id, env_kwargs_data['env_kwargs_val'], val_steps=train_config.val_steps),
            make('test', "eval"))


def plot_results(env_config, stats_dict_test: dict, str_id: str) -> None:
    """Three stacked panels of the test roll-out (prices, operating state + CH4 flow, reward + cumulative reward)
    saved as plots/<str_id>_plot.png (mctsq_utils.py:577-642).  The last six steps of an episode are the
    environment's data overhead and are not drawn; rewards are stored in ct and shown in EUR."""
    import matplotlib
    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt
    print("---Plot and save RL performance on the test set under ./plots/ ...\n")
    st, cut = stats_dict_test, slice(None, -6)
    days = (st['steps_stats'] * env_config.sim_step / 86400.0)[cut]
    fig, axs = plt.subplots(3, 1, figsize=(10, 6), sharex=True, sharey=False)

    def line(ax, key, label, colour=None, scale=1.0, offset=0.0, ylabel=None, loc="upper left"):
        ax.plot(days, st[key][cut] * scale + offset, *([colour] if colour else []), label=label)
        if ylabel is not None:
            ax.set_ylabel(ylabel)
        ax.legend(loc=loc, fontsize='small')

    line(axs[0], 'el_price_stats', 'El.')
    line(axs[0], 'gas_price_stats', '(S)NG', 'g', ylabel='El. & (S)NG prices\n [ct/kWh]')
    line(axs[0].twinx(), 'eua_price_stats', 'EUA', 'k', ylabel='EUA prices [€/t$_{CO2}$]', loc="upper right")
    line(axs[1], 'Meth_State_stats', 'state', 'b', offset=1.0, ylabel=' ')
    axs[1].set_yticks([1, 2, 3, 4, 5])
    axs[1].set_yticklabels(['Standby', 'Cooldown/Off', 'Startup', 'Partial Load', 'Full Load'])
    axs[1].grid(axis='y', linestyle='dashed')
    line(axs[1].twinx(), 'Meth_CH4_flow_stats', 'CH$_4$', 'yellowgreen', scale=1000.0,
         ylabel='CH$_4$ flow rate\n [mmol/s]', loc="upper right")
    line(axs[2], 'Meth_reward_stats', 'Reward', 'g', scale=0.01, ylabel='Reward [€]')
    axs[2].set_xlabel('Time [d]')
    line(axs[2].twinx(), 'Meth_cum_reward_stats', 'Cum. Reward', 'k', scale=0.01, ylabel='Cumulative \n reward [€]',
         loc="upper right")
    fig.suptitle(f"{str_id} \n Rew: {np.round(st['Meth_cum_reward_stats'][-7] / 100, 0)} €", fontsize=9)
    plt.savefig(f'plots/{str_id}_plot.png')
    plt.close()
