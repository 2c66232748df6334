#!/usr/bin/env python3
"""Golden vectors for calculate_optimum (dev container only): imports the reference's src/mctsq_opt.py
with cwd = /root/reference (it reads config/config_env.yaml relative to the cwd), runs it on seeded
synthetic price series and stores inputs, outputs and the configuration VALUES the function reads.
    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_opt.py"""
import contextlib, io, json, os, sys
import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
os.chdir(REF)
from src.mctsq_opt import calculate_optimum          # noqa: E402
from src.mctsq_config_env import EnvConfiguration    # noqa: E402

cfg = EnvConfiguration()
KEYS = ["scenario", "operation", "convert_mol_to_Nm3", "H_u_CH4", "H_u_H2", "eta_CHP", "eeg_el_price", "dt_water",
        "cp_water", "h_H2O_evap", "heat_price", "o2_price", "Molar_mass_CO2", "Molar_mass_H2O",
        "min_load_electrolyzer", "rho_water", "water_price", "price_ahead"]
values = {k: getattr(cfg, k) for k in KEYS}
values["meth_stats_load"] = cfg.meth_stats_load
rng = np.random.RandomState(3654)
cases = {}
for name, hours, days in (("toy48", 48, 2), ("edge", 49, 2), ("year", 24 * 40, 40), ("level", 30, 2)):
    el = rng.uniform(-20.0, 250.0, hours)          # ct/kWh-ish hourly prices incl. negative ones
    gas = rng.uniform(1.0, 12.0, days)
    eua = rng.uniform(20.0, 100.0, days)
    data_name = "reward_Level" if name == "level" else name
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res = calculate_optimum(el, gas, eua, data_name, cfg.stats_names)
    cases[name] = dict(el=el, gas=gas, eua=eua, printed=buf.getvalue())
    for k, v in res.items():
        cases[name]["out/" + k] = np.asarray(v)
flat = {}
for name, d in cases.items():
    for k, v in d.items():
        flat[f"{name}/{k}"] = np.asarray(v)
np.savez_compressed(os.path.join(OUT, "opt_cases.npz"), cfg=json.dumps(values), stats_names=json.dumps(cfg.stats_names), **flat)
print("wrote opt_cases.npz", len(flat), "arrays")
