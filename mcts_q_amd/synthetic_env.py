"""Synthetic stand-in for the `_per` PtG environment protocol (/root/reference/env/ptg_gym_env_per.py:422-431,
644-662): step([action, state_c]) is a pure function of its arguments and returns the successor in
info['state_c'].  The real environment needs gymnasium and the data CSVs (absent here); this one only has
the right SHAPES: a normalised observation dict of nine fp64 arrays (Elec_Price [25], six process traces
[10], Gas_Price / EUA_Price [3]) and a Meth_State in 0..4 that drives the legal-action mask."""
import numpy as np

PROC = ("T_CAT", "H2_in_MolarFlow", "CH4_syn_MolarFlow", "H2_res_MolarFlow", "H2O_DE_MassFlow", "Elec_Heating")
_NEXT_STATE = {0: 0, 1: 1, 2: 2, 3: 3, 4: 4}     # action a leads to operating state a


def _obs(seed):
    r = np.random.RandomState(seed % (2 ** 31 - 1))
    o = {"Elec_Price": r.random_sample(25)}
    for k in PROC:
        o[k] = r.random_sample(10)
    o["Gas_Price"], o["EUA_Price"] = r.random_sample(3), r.random_sample(3)
    return o


class FakePerEnv:
    def __init__(self, seed=0, horizon=1000):
        self.seed, self.horizon = seed, horizon

    def reset(self):
        h = (self.seed * 1000003 + 12345) & 0x7fffffff
        sc = {"obs_norm": _obs(h), "Meth_State": 0, "t": 0, "h": h, "cum": 0.0}
        return sc["obs_norm"], {"state_c": sc}

    def step(self, arg):
        action, sc = arg
        h = (sc["h"] * 6364136223846793005 + 1442695040888963407 * (int(action) + 1)) & 0x7fffffffffffffff
        h = (h >> 16) & 0x7fffffff
        obs = _obs(h)
        reward = float(((h >> 3) % 2001) / 1000.0 - 1.0)
        t = sc["t"] + 1
        nsc = {"obs_norm": obs, "Meth_State": _NEXT_STATE[int(action)], "t": t, "h": h, "cum": sc["cum"] + reward}
        terminated = t >= self.horizon
        info = {"state_c": nsc, "Meth_State": nsc["Meth_State"], "el_price_act": float(obs["Elec_Price"][12]),
                "Pot_Reward": reward, "cum_reward": nsc["cum"]}
        return obs, reward, terminated, False, info


class _Discrete:
    def __init__(self, n):
        self.n = n


class FakeGymEnv(FakePerEnv):
    """FakePerEnv with the two gym attributes MCTSQ reads (action_space.n, eps_sim_steps)."""

    def __init__(self, seed=0, horizon=1000, eps_sim_steps=8):
        super().__init__(seed, horizon)
        self.action_space = _Discrete(5)
        self.eps_sim_steps = eps_sim_steps


class FakeStdGymEnv:
    """Stateful stand-in for the non-`_per` environment protocol (/root/reference/env/ptg_gym_env.py: reset(),
    step(action), get_obs(), meth_state; deep-copied by the search, mctsq_config_mcts.py:168, 243): the same hash
    walk as FakePerEnv, kept on the object."""

    def __init__(self, seed=0, horizon=1000, eps_sim_steps=8):
        self._per = FakePerEnv(seed, horizon)
        self.action_space = _Discrete(5)
        self.eps_sim_steps = eps_sim_steps
        self.reset()

    def reset(self):
        obs, info = self._per.reset()
        self._sc = info["state_c"]
        return obs, {}

    @property
    def meth_state(self):
        return self._sc["Meth_State"]

    def get_obs(self):
        return self._sc["obs_norm"]

    def step(self, action):
        obs, reward, terminated, truncated, info = self._per.step([action, self._sc])
        self._sc = info.pop("state_c")
        return obs, reward, terminated, truncated, info


class FakeBatchEnv:
    """Vectorised synthetic environment for the batched-environment protocol of treesearch.LockstepMCTS:
    state vector = (hash, t, meth_state, cumulative reward); observations are a counter-based hash of the
    state, so step_batch is a pure function of (states, actions) like the `_per` environment."""
    state_dim = 4

    def __init__(self, horizon=1000):
        self.horizon = horizon
        self._lanes = (np.arange(91, dtype=np.uint64) + 1) * np.uint64(0x9E3779B97F4A7C15)

    def reset(self, seeds):
        seeds = np.asarray(seeds, dtype=np.uint64)
        st = np.zeros((len(seeds), 4))
        st[:, 0] = ((seeds * np.uint64(1000003) + np.uint64(12345)) & np.uint64(0x7fffffff)).astype(np.float64)
        return st

    def _rows(self, h):
        x = (h.astype(np.uint64)[:, None] + np.uint64(1)) * self._lanes[None, :]
        x ^= x >> np.uint64(29)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(32)
        return ((x >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24)))

    def observe(self, states):
        states = np.asarray(states)
        return self._rows(states[:, 0]), states[:, 2].astype(np.int64)

    def step_batch(self, states, actions):
        states = np.asarray(states)
        a = np.asarray(actions, dtype=np.uint64)
        h = states[:, 0].astype(np.uint64)
        h = (h * np.uint64(6364136223846793005) + np.uint64(1442695040888963407) * (a + np.uint64(1)))
        h = (h >> np.uint64(16)) & np.uint64(0x7fffffff)
        reward = ((h >> np.uint64(3)) % np.uint64(2001)).astype(np.float64) / 1000.0 - 1.0
        t = states[:, 1] + 1
        nxt = np.stack([h.astype(np.float64), t, a.astype(np.float64), states[:, 3] + reward], axis=1)
        term = t >= self.horizon
        return nxt, self._rows(h), reward, term, np.zeros(len(t), dtype=bool), a.astype(np.int64)


# ---- synthetic inputs for the REAL environment arithmetic (mcts_q_amd/ptgenv.py) --------------------------------
# The GPU box has neither the reference's CSV files nor its YAML; these builders make price series and process data
# sets of the real shapes (hourly / daily prices, 17 data sets of [rows, 7] at 2 s resolution) from a seed, with the
# plant constants of a typical config_env.yaml, and run them through the same Preprocessing as real data.
PTG_ENV_VALUES = dict(
    scenario=2, operation="OP2", price_ahead=13, price_past=12, time_step_op=2, sim_step=600, noise=10, eps_len_d=4,
    state_change_penalty=0.0,
    ptg_state_space=dict(standby=0, cooldown=1, startup=2, partial_load=3, full_load=4),
    meth_stats_load=dict(Meth_State=[2, 5, 5], Meth_Action=[6, 10, 11], Meth_Hot_Cold=[0, 1, 1],
                         Meth_T_cat=[11.0, 451.0, 451.0], Meth_H2_flow=[0.0, 0.0198, 0.0485],
                         Meth_CH4_flow=[0.0, 0.0048, 0.0114], Meth_H2_res_flow=[0.0, 0.000151, 0.0017],
                         Meth_H2O_flow=[0.0, 0.458545, 1.22], Meth_el_heating=[0.0, 350.0, 380.0]),
    ch4_price_fix=15.0, heat_price=4.6, o2_price=10.2, water_price=6.4, eeg_el_price=17.84, H_u_CH4=35.883,
    H_u_H2=10.783, h_H2O_evap=2257, dt_water=90, cp_water=4.18, rho_water=998, convert_mol_to_Nm3=0.02241407,
    Molar_mass_CO2=44.01, Molar_mass_H2O=18.02, eta_CHP=0.38, min_load_electrolyzer=0.032,
    r_0_values=dict(el_price=[0], gas_price=[10], eua_price=[50]),
    t_cat_standby=188.2, t_cat_startup_cold=160, t_cat_startup_hot=350,
    time1_start_p_f=1201, time2_start_f_p=151, time_p_f=210, time_f_p=126, time1_p_f_p=51, time2_p_f_p=151,
    time23_p_f_p=225, time3_p_f_p=301, time34_p_f_p=376, time4_p_f_p=451, time45_p_f_p=563, time5_p_f_p=675,
    time1_f_p_f=51, time2_f_p_f=151, time23_f_p_f=225, time3_f_p_f=301, time34_f_p_f=376, time4_f_p_f=451,
    time45_f_p_f=526, time5_f_p_f=601, i_fully_developed=12000, j_fully_developed=100,
    el_l_b=-10, el_u_b=90, gas_l_b=0.4, gas_u_b=32, eua_l_b=23, eua_u_b=98, T_l_b=10, T_u_b=600, h2_l_b=0, ch4_l_b=0,
    h2_res_l_b=0, h2o_l_b=0, heat_l_b=0, heat_u_b=1800)
PTG_TRAIN_VALUES = dict(com_conf="pc", device="auto", str_inv="synthetic", model_conf="simple_train", train_steps=100000,
                        val_steps=1000, r_seed_train=[3654], r_seed_test=[605])
PTG_MCTSQ_VALUES = dict(seq_length=10, seq_step=30)
_DATASET_ROWS = dict(startup_cold=3422, startup_hot=2048, cooldown=45001, standby_down=13829, standby_up=15546,
                     op1_start_p=4987, op2_start_f=2223, op3_p_f=3015, op4_p_f_p_5=2780, op5_p_f_p_10=3021,
                     op6_p_f_p_15=3261, op7_p_f_p_22=3481, op8_f_p=2820, op9_f_p_f_5=2947, op10_f_p_f_10=3297,
                     op11_f_p_f_15=3466, op12_f_p_f_20=3616)


def _curve(rng, rows, t0, t1, rate):
    """T_cat trace from t0 towards t1 (first-order lag) at 0.1 K resolution, like the measured files."""
    x = np.arange(rows) / max(rows - 1, 1)
    t = t1 + (t0 - t1) * np.exp(-rate * x) + rng.normal(0, 0.15, rows).cumsum() * 0.02
    return np.round(t, 1)


def synthetic_ptg_tables(seed: int = 0, scale: float = 1.0) -> dict:
    """dict_op_data stand-in: 17 data sets [rows, 7] (time, T_cat, n_h2, n_ch4, n_h2_res, m_DE, Pel)."""
    rng = np.random.RandomState(seed)
    load = PTG_ENV_VALUES["meth_stats_load"]
    lvl = {k: (load["Meth_H2_flow"][i], load["Meth_CH4_flow"][i], load["Meth_H2_res_flow"][i], load["Meth_H2O_flow"][i],
               load["Meth_el_heating"][i]) for k, i in (("off", 0), ("part", 1), ("full", 2))}
    spec = dict(startup_cold=(16, 455, 4.0, "off", "part"), startup_hot=(300, 455, 5.0, "off", "part"),
                cooldown=(598.7, 11, 5.0, "off", "off"), standby_down=(560, 188.2, 6.0, "off", "off"),
                standby_up=(11, 188.2, 6.0, "off", "off"), op1_start_p=(455, 470, 3.0, "part", "part"),
                op2_start_f=(455, 545, 3.0, "part", "full"), op3_p_f=(470, 550, 4.0, "part", "full"),
                op4_p_f_p_5=(480, 470, 3.0, "full", "part"), op5_p_f_p_10=(500, 470, 3.0, "full", "part"),
                op6_p_f_p_15=(520, 470, 3.0, "full", "part"), op7_p_f_p_22=(540, 470, 3.0, "full", "part"),
                op8_f_p=(550, 470, 4.0, "full", "part"), op9_f_p_f_5=(540, 550, 3.0, "part", "full"),
                op10_f_p_f_10=(520, 550, 3.0, "part", "full"), op11_f_p_f_15=(500, 550, 3.0, "part", "full"),
                op12_f_p_f_20=(480, 550, 3.0, "part", "full"))
    out = {}
    for name, (t0, t1, rate, a, b) in spec.items():
        rows = max(int(_DATASET_ROWS[name] * scale), 320)
        x = np.arange(rows) / (rows - 1)
        mix = 1.0 / (1.0 + np.exp(-(x - 0.15) * 40.0))                       # load ramps early in the file
        cols = [np.arange(rows) * 2.0, _curve(rng, rows, t0, t1, rate)]
        for q in range(5):
            v = lvl[a][q] * (1 - mix) + lvl[b][q] * mix
            cols.append(np.round(v * (1 + rng.normal(0, 0.01, rows)), 6 if q < 4 else 0))
        out[name] = np.stack(cols, axis=1)
    return out


def synthetic_ptg_prices(seed: int = 0, train_days: int = 14, eval_days: int = 6) -> dict:
    """dict_price_data stand-in before the scenario rule: hourly electricity, daily gas / EUA per split."""
    rng = np.random.RandomState(seed + 1)
    d = {}
    for split, days in (("train", train_days), ("val", eval_days), ("test", eval_days)):
        h = np.arange(days * 24)
        el = 9 + 6 * np.sin(2 * np.pi * h / 24) + 3 * np.sin(2 * np.pi * h / (24 * 7)) + rng.normal(0, 4, len(h))
        d[f"el_price_{split}"] = np.round(el, 3)
        d[f"gas_price_{split}"] = np.round(6 + rng.normal(0, 0.5, days).cumsum() * 0.3, 3)
        d[f"eua_price_{split}"] = np.round(70 + rng.normal(0, 1.0, days).cumsum(), 2)
    return d


def synthetic_ptg_kwargs(seed: int = 0, split: str = "train", scale: float = 1.0, train_days: int = 14,
                         env_values: dict | None = None, quiet: bool = True) -> dict:
    """The environment's dict_input built from synthetic data through the real Preprocessing."""
    import contextlib, io
    from .envconfig import EnvConfiguration
    from .ptgdata import Preprocessing, TrainConfiguration, apply_scenario, check_training_length
    ev = dict(PTG_ENV_VALUES)
    ev.update(env_values or {})
    ec = EnvConfiguration(values=ev)
    tc = TrainConfiguration(values=dict(PTG_TRAIN_VALUES), mctsq_values=PTG_MCTSQ_VALUES, env_values=ev)
    tc.seed_train, tc.seed_test = tc.r_seed_train[0], tc.r_seed_test[0]
    prices = synthetic_ptg_prices(seed, train_days)
    apply_scenario(ec, prices)
    prices.update({f"{k}_reward_level": ec.r_0_values[k] for k in ("el_price", "gas_price", "eua_price")})
    check_training_length(ec, len(prices["gas_price_train"]))
    with contextlib.redirect_stdout(io.StringIO() if quiet else None):
        pre = Preprocessing(prices, synthetic_ptg_tables(seed, scale), ec, tc)
    return pre.dict_env_kwargs(split)
