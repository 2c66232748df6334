"""Theoretical optimum T-OPT: host-side restatement of calculate_optimum
(/root/reference/src/mctsq_opt.py:33-196), vectorised over the hourly price series.

Start-up preprocessing (called four times by the reference's Preprocessing, src/mctsq_utils.py:263-278),
not part of the GPU hot path; it is here because the north star keeps this module's Python API.  The
arithmetic follows the reference's operation order term by term in float64, so results are bit-identical
(tests/test_opt_golden.py): everything that depends only on the load level is evaluated with Python
floats exactly as written there; the price-dependent products are element-wise numpy float64 ops, which
round like the scalar ones."""
import math

import numpy as np

from .envconfig import EnvConfiguration


def _level_terms(cfg, l):
    """Price-independent quantities of load level l (0 partial, 1 full): mctsq_opt.py:74-139."""
    m = cfg.meth_stats_load
    ch4_volumeflow = m['Meth_CH4_flow'][l + 1] * cfg.convert_mol_to_Nm3
    h2_res_volumeflow = m['Meth_H2_res_flow'][l + 1] * cfg.convert_mol_to_Nm3
    q_ch4 = ch4_volumeflow * cfg.H_u_CH4 * 1000
    q_h2_res = h2_res_volumeflow * cfg.H_u_H2 * 1000
    b_s3 = 1 if cfg.scenario == 3 else 0
    power_chp = q_ch4 * cfg.eta_CHP * b_s3
    q_chp = q_ch4 * (1 - cfg.eta_CHP) * b_s3
    chp_revenues = power_chp * cfg.eeg_el_price
    q_steam = (m['Meth_H2O_flow'][l + 1] * (cfg.dt_water * cfg.cp_water + cfg.h_H2O_evap) / 3600)
    steam_revenues = (q_steam + q_chp) * cfg.heat_price
    h2_volumeflow = m['Meth_H2_flow'][l + 1] * cfg.convert_mol_to_Nm3
    o2_volumeflow = 1 / 2 * h2_volumeflow * 3600
    o2_revenues = o2_volumeflow * cfg.o2_price
    meth_co2_mass_flow = m['Meth_CH4_flow'][l + 1] * cfg.Molar_mass_CO2 / 1000
    load_elec = h2_volumeflow / cfg.max_h2_volumeflow
    if load_elec < cfg.min_load_electrolyzer:
        eta = 0.02
    else:
        eta = (0.598 - 0.325 * load_elec ** 2 + 0.218 * load_elec ** 3 + 0.01 * load_elec ** (-1) -
               1.68 * 10 ** (-3) * load_elec ** (-2) + 2.51 * 10 ** (-5) * load_elec ** (-3))
    water_elec = m['Meth_H2_flow'][l + 1] * cfg.Molar_mass_H2O / 1000 * 3600
    water_costs = ((m['Meth_H2O_flow'][l + 1] + water_elec) / cfg.rho_water * cfg.water_price)
    return dict(gas_kw=q_ch4 + q_h2_res, chp=chp_revenues, steam=steam_revenues, o2=o2_revenues,
                eua_k=meth_co2_mass_flow / 1000 * 3600, heat_k=m['Meth_el_heating'][l + 1] / 1000,
                elec_k=h2_volumeflow * cfg.H_u_H2 * 1000 / eta, water=water_costs)


def calculate_optimum(el_price_data, gas_price_data, eua_price_data, data_name, stats_names, env_config=None):
    """Same signature and return value as the reference (mctsq_opt.py:33-39); `env_config` (extra,
    optional) supplies an already loaded configuration instead of reading config/config_env.yaml."""
    cfg = env_config if env_config is not None else EnvConfiguration()
    m = cfg.meth_stats_load
    el = np.asarray(el_price_data, dtype=np.float64)
    gas = np.asarray(gas_price_data, dtype=np.float64)
    eua = np.asarray(eua_price_data, dtype=np.float64)
    T = len(el)
    stats = np.zeros((T, len(stats_names)))
    if T:
        t_day = np.arange(T) // 24                              # math.floor(t / 24)
        t_day = np.where(t_day == len(gas), t_day - 1, t_day)   # mctsq_opt.py:67-68
        gas_h, eua_h = gas[t_day], eua[t_day]
        rew, parts = [], []
        for l in range(2):
            k = _level_terms(cfg, l)
            ch4_rev = k['gas_kw'] * gas_h
            eua_rev = k['eua_k'] * eua_h * 100
            heat = k['heat_k'] * el
            elec = k['elec_k'] * el
            elec_costs = heat + elec
            r = (ch4_rev + k['chp'] + k['steam'] + eua_rev + k['o2'] - elec_costs - k['water'])
            rew.append(r)
            parts.append((ch4_rev, np.full(T, k['steam']), np.full(T, k['o2']), eua_rev, np.full(T, k['chp']),
                          -heat, -elec, np.full(T, -k['water'])))
        index = np.where(rew[1] > rew[0], 1, 0)                 # list.index(max): first maximum wins ties
        best = np.where(index == 1, rew[1], rew[0])
        on = best > 0
        stats[:, 0] = np.arange(T)
        stats[:, 1], stats[:, 2], stats[:, 3] = el, gas_h, eua_h
        col = np.where(on, index + 1, 0)                        # row of meth_stats_load to report
        for c, key in enumerate(('Meth_State', 'Meth_Action', 'Meth_Hot_Cold', 'Meth_T_cat', 'Meth_H2_flow',
                                 'Meth_CH4_flow', 'Meth_H2O_flow', 'Meth_el_heating')):
            stats[:, 4 + c] = np.asarray(m[key], dtype=np.float64)[col]
        # the revenue / cost columns report the values of the LAST level evaluated (full load): the
        # reference reads the loop variables after the loop (mctsq_opt.py:164-165), whichever level won
        for c in range(8):
            stats[:, 12 + c] = np.where(on, parts[1][c], 0.0)
        stats[:, 23] = np.where(on, index, -1)
        stats[:, 20] = best
        stats[:, 21] = np.cumsum(np.where(on, best, 0.0))       # sequential left-to-right sum, like cum_rew +=
    out = {name: stats[:, i] for i, name in enumerate(stats_names)}
    if data_name != "reward_Level":
        max_pot_cum_rew = out['Meth_cum_reward_stats'][-cfg.price_ahead]
        print("    > ", data_name, ": Cumulative reward - theoretical optimum T-OPT = ", round(max_pot_cum_rew, 2))
    return out
