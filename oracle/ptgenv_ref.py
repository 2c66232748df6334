"""TEST INFRASTRUCTURE - scalar CPU restatement of the `_per` PtG environment, one state at a time.

Follows /root/reference/env/ptg_gym_env_per.py: reset (:664-705 with :102-195), step (:422-662), reward (:342-420),
_get_index (:715-725), _perform_sim_step (:727-773), the five mode changes (:797-992).  Like the reference it
materialises the rows of one simulation step as an array (slices, np.concatenate, np.ones * last row), takes
np.average over it, picks the look-back window by reversed strided slicing and finds rows by argmin.  The object keeps
hot_cold, the gas/EUA window and cum_rew between calls exactly as PTGEnv does, so it reproduces the reference on
SEQUENTIAL episodes; the product (mcts_q_amd/csrc_host/ptg_env.c) carries that state in the state vector.

PARITY UNPINNED: the reference environment imports gymnasium, which is absent in this image, so it cannot be run to
pin this restatement; the reference holds no fixtures for the environment either.  Only tests/ may import this file.

The one random draw of a mode change (np_random.normal(0, noise), :811, 825, 849) is passed in (`nz`).
"""
import numpy as np

DS = ("startup_cold", "startup_hot", "cooldown", "standby_down", "standby_up", "op1_start_p", "op2_start_f", "op3_p_f",
      "op4_p_f_p_5", "op5_p_f_p_10", "op6_p_f_p_15", "op7_p_f_p_22", "op8_f_p", "op9_f_p_f_5", "op10_f_p_f_10",
      "op11_f_p_f_15", "op12_f_p_f_20")
ID = {n: i for i, n in enumerate(DS)}
STANDBY, COOLDOWN, STARTUP, PARTIAL, FULL = range(5)


class RefPtGEnv:
    def __init__(self, kw):
        self.c = dict(kw)
        self.tab = [np.asarray(kw[n], dtype=np.float64) for n in DS]
        self.step_size = int(kw["sim_step"] / kw["time_step_op"])
        self.b_s3 = 1 if kw["scenario"] == 3 else 0
        self.d0 = 0

    # -- helpers -------------------------------------------------------------------------------------------------
    def _idx(self, ds, t_cat):
        return int(np.abs(self.tab[ds][:, 1] - t_cat).argmin())

    def _rows(self, a, init_state, b, next_state, i, j, change):
        """-> (rows [step, 7], state, i, j)"""
        A, S = self.tab[a], self.step_size
        total = len(A)
        if i + j * S < total:
            return A[i + (j - 1) * S:i + j * S], init_state, i, j
        over = i + j * S - total
        if over >= S:
            return np.ones((S, 7)) * A[-1], next_state, i, j
        head = A[i + (j - 1) * S:]
        if change:
            return np.concatenate((head, self.tab[b][:over]), axis=0), next_state, over, 0
        return np.concatenate((head, np.ones((over, 7)) * A[-1]), axis=0), next_state, i, j

    def _norm(self, seq):
        c = self.c
        lb = [c["T_l_b"], c["h2_l_b"], c["ch4_l_b"], c["h2_res_l_b"], c["h2o_l_b"], c["heat_l_b"]]
        ub = [c["T_u_b"], c["h2_u_b"], c["ch4_u_b"], c["h2_res_u_b"], c["h2o_u_b"], c["heat_u_b"]]
        return [(seq[:, q + 1] - lb[q]) / (ub[q] - lb[q]) for q in range(6)]

    def _obs(self):
        c = self.c
        keys = ("T_CAT", "H2_in_MolarFlow", "CH4_syn_MolarFlow", "H2_res_MolarFlow", "H2O_DE_MassFlow", "Elec_Heating")
        o = {"Elec_Price": (self.e_act[0] - c["el_l_b"]) / (c["el_u_b"] - c["el_l_b"]),
             "Gas_Price": (self.ge_d[0] - c["gas_l_b"]) / (c["gas_u_b"] - c["gas_l_b"]),
             "EUA_Price": (self.ge_d[1] - c["eua_l_b"]) / (c["eua_u_b"] - c["eua_l_b"])}
        o.update(dict(zip(keys, self._norm(self.seq))))
        return {k: np.array(v, dtype=np.float64) for k, v in o.items()}

    def _state_c(self, obs):
        return dict(i=self.i, j=self.j, k=self.k, Meth_State=self.m, Meth_T_cat=self.t_cat, standby_op=self.sb,
                    startup_op=self.su, part_op=self.pt, full_op=self.fl, obs_norm=obs)

    # -- reset ---------------------------------------------------------------------------------------------------
    def reset(self, d0=0):
        c = self.c
        self.d0 = int(d0)
        self.e_act = c["e_r_b"][:, :, self.d0 * 24]
        self.g_act = c["g_e"][:, :, self.d0]
        self.ge_d = np.zeros((2, 3))
        self.ge_d[:, 0] = c["g_e"][:, 0, self.d0 - 1]
        self.ge_d[:, 1] = c["g_e"][:, 0, self.d0]
        self.ge_d[:, 2] = c["g_e"][:, 0, self.d0]
        self.m, self.sb, self.su, self.pt, self.fl = COOLDOWN, ID["standby_down"], ID["startup_cold"], ID["op1_start_p"], ID["op2_start_f"]
        self.t_cat = 16
        self.i, self.j, self.k = self._idx(ID["cooldown"], self.t_cat), 0, 0
        self.seq = np.ones((c["seq_length"], 7)) * self.tab[ID["cooldown"]][self.i]
        self.hot, self.cum = 0, 0
        obs = self._obs()
        return obs, {"state_c": self._state_c(obs), "cum_reward": self.cum}

    # -- step ----------------------------------------------------------------------------------------------------
    def step(self, action, sc, nz=0.0):
        c, S = self.c, self.step_size
        self.i, self.j, self.k, self.m, self.t_cat = sc["i"], sc["j"], sc["k"], sc["Meth_State"], sc["Meth_T_cat"]
        self.sb, self.su, self.pt, self.fl = sc["standby_op"], sc["startup_op"], sc["part_op"], sc["full_op"]
        k = self.k
        if self.t_cat <= c["t_cat_startup_cold"]:
            self.hot = 0
        elif self.t_cat >= c["t_cat_startup_hot"]:
            self.hot = 1
        m = self.m
        stay = {(0, STANDBY), (1, COOLDOWN), (2, STARTUP), (2, PARTIAL), (2, FULL), (3, STANDBY), (3, COOLDOWN),
                (3, STARTUP), (3, PARTIAL), (4, STANDBY), (4, COOLDOWN), (4, STARTUP), (4, FULL)}
        if (action, m) in stay:
            self.j += 1
            if m == STANDBY:
                args = (self.sb, m, self.sb, m, False)
            elif m == COOLDOWN:
                args = (ID["cooldown"], m, ID["cooldown"], m, False)
            elif m == STARTUP:
                args = (self.su, m, self.pt, PARTIAL, True)
            elif m == PARTIAL:
                args = (self.pt, m, self.pt, PARTIAL, False)
            else:
                args = (self.fl, m, self.fl, FULL, False)
        elif action == 0:
            self.m = STANDBY
            self.sb = ID["standby_up"] if self.t_cat <= c["t_cat_standby"] else ID["standby_down"]
            self.i, self.j = int(max(self._idx(self.sb, self.t_cat) + nz, 0)), 1
            args = (self.sb, self.m, self.sb, self.m, False)
        elif action == 1:
            self.m = COOLDOWN
            self.i, self.j = int(max(self._idx(ID["cooldown"], self.t_cat) + nz, 0)), 1
            args = (ID["cooldown"], self.m, ID["cooldown"], self.m, False)
        elif action == 2:
            self.m = STARTUP
            self.pt, self.fl = ID["op1_start_p"], ID["op2_start_f"]
            self.su = ID["startup_cold"] if self.hot == 0 else ID["startup_hot"]
            self.i, self.j = int(max(self._idx(self.su, self.t_cat) + nz, 0)), 1
            args = (self.su, self.m, self.pt, PARTIAL, True)
        elif action == 3:
            self._to_partial()
            args = (self.pt, self.m, self.pt, PARTIAL, False)
        else:
            self._to_full()
            args = (self.fl, self.m, self.fl, FULL, False)
        op, self.m, self.i, self.j = self._rows(args[0], args[1], args[2], args[3], self.i, self.j, args[4])

        hours = (k + 1) * c["sim_step"] / 3600
        days = hours / 24
        self.e_act = c["e_r_b"][:, :, self.d0 * 24 + int(np.floor(hours))]
        self.g_act = c["g_e"][:, :, self.d0 + int(np.floor(days))]
        self.ge_d[:, 1] = self.g_act[:, 0]
        if days % 0.5 == 0 and days % 1 != 0:
            self.ge_d[:, 0] = self.ge_d[:, 1]
            self.ge_d[:, 2] = self.g_act[:, 1]
        self.t_cat = op[-1, 1]
        flows = [np.average(op[:, q]) for q in range(2, 7)]
        self.seq = np.array([op[::-1, q][::c["seq_step"]][:c["seq_length"]] for q in range(7)]).T
        reward, rew = self._reward(*flows)
        obs = self._obs()
        terminated = self.k == c["eps_sim_steps"] - 6
        self.k += 1
        info = {"state_c": self._state_c(obs), "el_price_act": self.e_act[0, 12], "gas_price_act": self.g_act[0, 0],
                "eua_price_act": self.g_act[1, 0], "Pot_Reward": self.e_act[1, 12], "Part_Full": self.e_act[2, 12],
                "cum_reward": self.cum, "Meth_State": self.m, "Meth_T_cat": self.t_cat, "Meth_Hot_Cold": self.hot,
                "reward [ct]": rew}
        return obs, reward, terminated, False, info

    def _to_partial(self):
        c = self.c
        self.m = PARTIAL
        t = self.i + self.j * self.step_size
        nxt = ("op8_f_p", 0, 1)
        if self.fl == ID["op2_start_f"]:
            if t < c["time2_start_f_p"]:
                nxt = ("op1_start_p", self._idx(ID["op1_start_p"], self.t_cat), 1)
        elif self.fl == ID["op3_p_f"]:
            if t < c["time1_p_f_p"]:
                nxt = ("op8_f_p", c["i_fully_developed"], c["j_fully_developed"])
            elif c["time1_p_f_p"] < t < c["time2_p_f_p"]:
                nxt = ("op4_p_f_p_5", self.i, self.j + 1)
            elif c["time2_p_f_p"] < t < c["time_p_f"]:
                nxt = ("op4_p_f_p_5", c["time2_p_f_p"], 1)
            elif c["time_p_f"] < t < c["time34_p_f_p"]:
                nxt = ("op5_p_f_p_10", c["time3_p_f_p"], 1)
            elif c["time34_p_f_p"] < t < c["time45_p_f_p"]:
                nxt = ("op6_p_f_p_15", c["time4_p_f_p"], 1)
            elif c["time45_p_f_p"] < t < c["time5_p_f_p"]:
                nxt = ("op7_p_f_p_22", c["time5_p_f_p"], 1)
        self.pt, self.i, self.j = ID[nxt[0]], nxt[1], nxt[2]

    def _to_full(self):
        c = self.c
        self.m = FULL
        t = self.i + self.j * self.step_size
        nxt = ("op3_p_f", 0, 1)
        if self.pt == ID["op1_start_p"]:
            if t < c["time1_start_p_f"]:
                nxt = ("op2_start_f", 0, 1)
        elif self.pt == ID["op8_f_p"]:
            if t < c["time1_f_p_f"]:
                nxt = ("op3_p_f", c["i_fully_developed"], c["j_fully_developed"])
            elif c["time1_f_p_f"] < t < c["time_f_p"]:
                nxt = ("op9_f_p_f_5", self.i, self.j + 1)
            elif c["time_f_p"] < t < c["time23_f_p_f"]:
                nxt = ("op9_f_p_f_5", c["time2_f_p_f"], 1)
            elif c["time23_f_p_f"] < t < c["time34_f_p_f"]:
                nxt = ("op10_f_p_f_10", c["time3_f_p_f"], 1)
            elif c["time34_f_p_f"] < t < c["time45_f_p_f"]:
                nxt = ("op11_f_p_f_15", c["time4_f_p_f"], 1)
            elif c["time45_f_p_f"] < t < c["time5_f_p_f"]:
                nxt = ("op12_f_p_f_20", c["time5_f_p_f"], 1)
        self.fl, self.i, self.j = ID[nxt[0]], nxt[1], nxt[2]

    def _reward(self, h2, ch4, h2_res, h2o, heating):
        c = self.c
        gas, eua, el = self.g_act[0, 0], self.g_act[1, 0], self.e_act[0, 12]
        q_ch4 = ch4 * c["convert_mol_to_Nm3"] * c["H_u_CH4"] * 1000
        q_h2_res = h2_res * c["convert_mol_to_Nm3"] * c["H_u_H2"] * 1000
        ch4_rev = (q_ch4 + q_h2_res) * gas
        chp_rev = q_ch4 * c["eta_CHP"] * self.b_s3 * c["eeg_el_price"]
        q_chp = q_ch4 * (1 - c["eta_CHP"]) * self.b_s3
        q_steam = h2o * (c["dt_water"] * c["cp_water"] + c["h_H2O_evap"]) / 3600
        steam_rev = (q_steam + q_chp) * c["heat_price"]
        h2_vol = h2 * c["convert_mol_to_Nm3"]
        o2_rev = 1 / 2 * h2_vol * 3600 * c["o2_price"]
        eua_rev = ch4 * c["Molar_mass_CO2"] / 1000 / 1000 * 3600 * eua * 100
        load = h2_vol / c["max_h2_volumeflow"]
        if load < c["min_load_electrolyzer"]:
            eta = 0.02
        else:
            eta = (0.598 - 0.325 * load ** 2 + 0.218 * load ** 3 + 0.01 * load ** (-1) - 1.68 * 10 ** (-3) * load ** (-2) +
                   2.51 * 10 ** (-5) * load ** (-3))
        costs_el = heating / 1000 * el + h2_vol * c["H_u_H2"] * 1000 / eta * el
        water = (h2o + h2 * c["Molar_mass_H2O"] / 1000 * 3600) / c["rho_water"] * c["water_price"]
        rew = (ch4_rev + chp_rev + steam_rev + eua_rev + o2_rev - costs_el - water) * c["sim_step"] / 3600
        self.cum += rew
        return (rew - c["rew_l_b"]) / (c["rew_u_b"] - c["rew_l_b"]), rew
