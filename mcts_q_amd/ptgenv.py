"""Vectorised PtG environment: PTGEnv.step of the `_per` environment over many states at once.

The reference steps ONE state per Python call (/root/reference/env/ptg_gym_env_per.py:422-662); after the lock-step
driver batches the network calls, that call is the host bottleneck.  `PtGBatchEnv` takes the same `dict_input` the
reference environment is built from (Preprocessing.dict_env_kwargs, src/mctsq_utils.py:419-493 - see
mcts_q_amd/ptgdata.py) and implements the batched-environment protocol of treesearch.LockstepMCTS on state vectors;
the arithmetic is in libmctsq_tree.so (mcts_q_amd/csrc_host/ptg_env.c; layout and deviations in include/mctsq_tree.h).
There is no Python fallback: without the library the constructor raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

DATASETS = ("startup_cold", "startup_hot", "cooldown", "standby_down", "standby_up", "op1_start_p", "op2_start_f",
            "op3_p_f", "op4_p_f_p_5", "op5_p_f_p_10", "op6_p_f_p_15", "op7_p_f_p_22", "op8_f_p", "op9_f_p_f_5",
            "op10_f_p_f_10", "op11_f_p_f_15", "op12_f_p_f_20")          # load_data's order (mctsq_utils.py:126-134)
STATE_HEAD = 12
_DOUBLES = ("convert_mol_to_Nm3", "H_u_CH4", "H_u_H2", "dt_water", "cp_water", "rho_water", "Molar_mass_CO2",
            "Molar_mass_H2O", "h_H2O_evap", "eeg_el_price", "heat_price", "o2_price", "water_price",
            "min_load_electrolyzer", "max_h2_volumeflow", "eta_CHP", "t_cat_standby", "t_cat_startup_cold",
            "t_cat_startup_hot", "time1_start_p_f", "time2_start_f_p", "time_p_f", "time_f_p", "time1_p_f_p",
            "time2_p_f_p", "time3_p_f_p", "time34_p_f_p", "time4_p_f_p", "time45_p_f_p", "time5_p_f_p", "time1_f_p_f",
            "time2_f_p_f", "time23_f_p_f", "time3_f_p_f", "time34_f_p_f", "time4_f_p_f", "time45_f_p_f", "time5_f_p_f",
            "i_fully_developed", "j_fully_developed")
_BOUNDS = ("el", "gas", "eua", "T", "h2", "ch4", "h2_res", "h2o", "heat")


class PtgParams(C.Structure):
    """mctsq_ptg_params (include/mctsq_tree.h)."""
    _fields_ = ([(k, C.c_int64) for k in ("step_size", "seq_length", "seq_step", "sim_step", "eps_sim_steps", "price_len",
                                          "H", "Dd", "scenario3")] +
                [(k, C.c_double) for k in _DOUBLES] +
                [("lb", C.c_double * 9), ("ub", C.c_double * 9), ("rew_l_b", C.c_double), ("rew_u_b", C.c_double)])


_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("MCTSQ_TREE_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmctsq_tree.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with make -C mcts_q_amd/csrc")
        L = C.CDLL(path)
        P, I64 = C.c_void_p, C.c_int64
        L.mctsq_ptg_create.argtypes = [C.POINTER(PtgParams), P, P, P, P]
        L.mctsq_ptg_create.restype = P
        L.mctsq_ptg_destroy.argtypes = [P]
        L.mctsq_ptg_destroy.restype = None
        L.mctsq_ptg_reset.argtypes = [P, I64, P, P]
        L.mctsq_ptg_reset.restype = None
        L.mctsq_ptg_observe.argtypes = [P, I64, P, P, I64, P, I64, P, I64, P]
        L.mctsq_ptg_observe.restype = I64
        L.mctsq_ptg_step.argtypes = [P, I64, P, P, P, P, P, I64, P, I64, P, I64, P, P, P]
        L.mctsq_ptg_step.restype = I64
        L.mctsq_ptg_info.argtypes = [P, I64, P, P]
        L.mctsq_ptg_info.restype = I64
        L.mctsq_ptg_state_dim.argtypes = [P]
        L.mctsq_ptg_state_dim.restype = I64
        _LIB = L
    return _LIB


class PtGBatchEnv:
    """Batched-environment protocol (treesearch.LockstepMCTS): state_dim, reset, observe, step_batch.

    dict_input: the reference's environment kwargs (same keys).  `eps_ind` (training) maps episode numbers to day
    offsets as PTGEnv.__init__/reset do (:74-79, 675-680); without it every episode starts at day 0.
    seed: generator of the mode-change index noise N(0, noise) (the reference draws it from gym's np_random)."""

    INFO_KEYS = ("el_price_act", "gas_price_act", "eua_price_act", "Pot_Reward", "Part_Full", "cum_reward", "Meth_T_cat",
                 "Meth_Hot_Cold")

    def __init__(self, dict_input: dict, seed: int = 0):
        kw = dict_input
        if kw.get("action_type", "discrete") != "discrete":
            raise ValueError("PtGBatchEnv implements the discrete action space only")
        L = self._L = _lib()
        p = PtgParams()
        sim_step, ts = kw["sim_step"], kw["time_step_op"]
        if int(sim_step) != sim_step:
            raise ValueError("sim_step must be a whole number of seconds")
        p.step_size, p.sim_step = int(sim_step / ts), int(sim_step)
        p.seq_length, p.seq_step, p.eps_sim_steps = int(kw["seq_length"]), int(kw["seq_step"]), int(kw["eps_sim_steps"])
        p.price_len = int(kw["price_ahead"] + kw["price_past"])
        e_r_b = np.ascontiguousarray(kw["e_r_b"], dtype=np.float64)
        g_e = np.ascontiguousarray(kw["g_e"], dtype=np.float64)
        if e_r_b.ndim != 3 or e_r_b.shape[:2] != (3, p.price_len) or g_e.ndim != 3 or g_e.shape[:2] != (2, 2):
            raise ValueError(f"e_r_b must be [3, {p.price_len}, H] and g_e [2, 2, D]")
        p.H, p.Dd, p.scenario3 = e_r_b.shape[2], g_e.shape[2], int(kw["scenario"] == 3)
        for k in _DOUBLES:
            setattr(p, k, float(kw[k]))
        for q, b in enumerate(_BOUNDS):
            p.lb[q], p.ub[q] = float(kw[f"{b}_l_b"]), float(kw[f"{b}_u_b"])
        p.rew_l_b, p.rew_u_b = float(kw["rew_l_b"]), float(kw["rew_u_b"])
        tabs = [np.ascontiguousarray(kw[n], dtype=np.float64) for n in DATASETS]
        if any(t.ndim != 2 or t.shape[1] != 7 for t in tabs):
            raise ValueError("process data sets must be [rows, 7] arrays (import_data, mctsq_utils.py:54-82)")
        op = np.ascontiguousarray(np.concatenate(tabs, axis=0))
        lens = np.array([len(t) for t in tabs], dtype=np.int64)
        self._h = L.mctsq_ptg_create(C.byref(p), op.ctypes.data, lens.ctypes.data, e_r_b.ctypes.data, g_e.ctypes.data)
        if not self._h:
            raise ValueError("mctsq_ptg_create refused the geometry: every data set needs at least one simulation step of "
                             "rows, (seq_length - 1) * seq_step must fit in one step, sim_step must divide 43200 s")
        self.params = p
        self.state_dim = int(L.mctsq_ptg_state_dim(self._h))
        self.row_width = p.price_len + 6 * p.seq_length + 6
        self.noise = float(kw["noise"])
        self.eps_sim_steps = p.eps_sim_steps
        eps_ind = kw.get("eps_ind")
        self.eps_ind = np.asarray(eps_ind).astype(np.int64) if isinstance(eps_ind, np.ndarray) else None
        self.eps_len_d = int(kw["eps_len_d"])
        self.rng = np.random.default_rng(seed)
        self.c_env = self._h                   # handle for the C lock-step driver (treesearch.py)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.mctsq_ptg_destroy(h)

    # ---- protocol ------------------------------------------------------------------------------
    def day_offsets(self, episodes):
        """act_ep_d of PTGEnv (:74-79): eps_ind[episode] * eps_len_d for training, 0 otherwise."""
        ep = np.asarray(episodes, dtype=np.int64)
        if self.eps_ind is None:
            return np.zeros(len(ep), dtype=np.int64)
        return self.eps_ind[ep % len(self.eps_ind)] * self.eps_len_d

    def reset(self, episodes):
        d0 = np.ascontiguousarray(self.day_offsets(episodes))
        st = np.empty((len(d0), self.state_dim))
        self._L.mctsq_ptg_reset(self._h, len(d0), d0.ctypes.data, st.ctypes.data)
        return st

    def _check(self, bad, what):
        if bad:
            raise IndexError(f"{what}: {bad} states are invalid or their clock points outside the price tables")

    def observe(self, states):
        st = np.ascontiguousarray(states, dtype=np.float64).reshape(-1, self.state_dim)
        n, w, P = len(st), self.row_width, self.params.price_len
        rows, meth = np.empty((n, w), dtype=np.float32), np.empty(n, dtype=np.int64)
        base = rows.ctypes.data
        self._check(self._L.mctsq_ptg_observe(self._h, n, st.ctypes.data, base, w, base + 4 * P, w,
                                              base + 4 * (w - 6), w, meth.ctypes.data), "observe")
        return rows, meth

    def step_batch(self, states, actions, noise=None):
        """-> next_states [n, D], rows fp32 [n, 91], reward [n], terminated [n], truncated [n], meth_state [n].
        noise: the N(0, noise) index offsets of mode changes, one per state (drawn here if None)."""
        st = np.ascontiguousarray(states, dtype=np.float64).reshape(-1, self.state_dim)
        n, w, P = len(st), self.row_width, self.params.price_len
        act = np.ascontiguousarray(actions, dtype=np.int64)
        if len(act) != n:
            raise ValueError("one action per state")
        nz = self.rng.normal(0.0, self.noise, n) if noise is None else np.ascontiguousarray(noise, dtype=np.float64)
        nxt, rows = np.empty_like(st), np.empty((n, w), dtype=np.float32)
        rew, term, meth = np.empty(n), np.empty(n, dtype=np.uint8), np.empty(n, dtype=np.int64)
        base = rows.ctypes.data
        self._check(self._L.mctsq_ptg_step(self._h, n, st.ctypes.data, act.ctypes.data, nz.ctypes.data, nxt.ctypes.data,
                                           base, w, base + 4 * P, w, base + 4 * (w - 6), w, rew.ctypes.data,
                                           term.ctypes.data, meth.ctypes.data), "step_batch")
        return nxt, rows, rew, term.astype(bool), np.zeros(n, dtype=bool), meth

    def info(self, states):
        """The evaluation fields of _get_info (:312-340) that are functions of a state, as a dict of arrays."""
        st = np.ascontiguousarray(states, dtype=np.float64).reshape(-1, self.state_dim)
        out = np.empty((len(st), 8))
        self._check(self._L.mctsq_ptg_info(self._h, len(st), st.ctypes.data, out.ctypes.data), "info")
        d = {k: out[:, q] for q, k in enumerate(self.INFO_KEYS)}
        d["Meth_State"] = st[:, 3].astype(np.int64)
        d["step"] = st[:, 2].astype(np.int64)
        return d

    # ---- state_c dicts of the reference <-> state vectors (interchange with the per-state protocol) ----------
    def state_from_state_c(self, sc: dict, day_offset: int = 0, hot_cold: int = 0, cum_reward: float = 0.0):
        """One state vector from a reference state_c dict (:644-655); op names or data-set ids are accepted."""
        ids = {n: i for i, n in enumerate(DATASETS)}
        st = np.zeros(self.state_dim)
        st[:5] = sc["i"], sc["j"], sc["k"], sc["Meth_State"], sc["Meth_T_cat"]
        for q, k in enumerate(("standby_op", "startup_op", "part_op", "full_op")):
            st[5 + q] = ids[sc[k]] if isinstance(sc[k], str) else int(sc[k])
        st[9:12] = hot_cold, day_offset, cum_reward
        o = sc["obs_norm"]
        keys = ("T_CAT", "H2_in_MolarFlow", "CH4_syn_MolarFlow", "H2_res_MolarFlow", "H2O_DE_MassFlow", "Elec_Heating")
        st[STATE_HEAD:] = np.stack([np.asarray(o[k], dtype=np.float64) for k in keys], axis=-1).ravel()
        return st
