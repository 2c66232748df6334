"""Environment configuration loader: host-side mirror of EnvConfiguration
(/root/reference/src/mctsq_config_env.py:15-73).  Start-up plumbing, not on the hot path: it exists so
that `src/mctsq_opt.calculate_optimum` and callers of the reference's module paths resolve here.

The YAML is the USER'S `config/config_env.yaml` (cwd-relative, as in the reference); nothing of it is
shipped in this repository."""
import yaml

STATS_NAMES = [
    'steps_stats', 'el_price_stats', 'gas_price_stats', 'eua_price_stats',
    'Meth_State_stats', 'Meth_Action_stats', 'Meth_Hot_Cold_stats', 'Meth_T_cat_stats',
    'Meth_H2_flow_stats', 'Meth_CH4_flow_stats', 'Meth_H2O_flow_stats', 'Meth_el_heating_stats',
    'Meth_ch4_revenues_stats', 'Meth_steam_revenues_stats', 'Meth_o2_revenues_stats',
    'Meth_eua_revenues_stats', 'Meth_chp_revenues_stats', 'Meth_elec_costs_heating_stats',
    'Meth_elec_costs_electrolyzer_stats', 'Meth_water_costs_stats', 'Meth_reward_stats',
    'Meth_cum_reward_stats', 'pot_reward_stats', 'part_full_stats',
]


class EnvConfiguration:
    """YAML keys become attributes; derived fields as in mctsq_config_env.py:24-73 (same names,
    same ValueErrors)."""

    def __init__(self, path: str = "config/config_env.yaml", values: dict | None = None):
        if values is None:
            with open(path, "r", encoding="utf-8") as f:
                values = yaml.safe_load(f)
        self.__dict__.update(values)
        if self.scenario not in [1, 2, 3]:
            raise ValueError(f"Specified business scenario ({self.scenario}) "
                             "must match one of the three implemented scenarios [1,2,3]!")
        self.train_len_d = None
        if self.operation not in ['OP1', 'OP2']:
            raise ValueError("Invalid load level specified - data/config_env.yaml -> operation "
                             f": {values['operation']} must match ['OP1', 'OP2']")
        if isinstance(getattr(self, "datafile_path", None), dict):
            base = self.datafile_path['path'] + self.operation
            for i in range(2, 19):
                setattr(self, f'datafile_path{i}', f"{base}/{self.datafile_path['datafile'][f'datafile_path{i}']}")
        if 'OP1' in self.meth_stats_load or 'OP2' in self.meth_stats_load:
            self.meth_stats_load = self.meth_stats_load[self.operation]
        load = self.meth_stats_load
        self.max_h2_volumeflow = self.convert_mol_to_Nm3 * load['Meth_H2_flow'][2]
        self.h2_u_b = load['Meth_H2_flow'][2]
        self.ch4_u_b = load['Meth_CH4_flow'][2]
        self.h2_res_u_b = load['Meth_H2_res_flow'][2]
        self.h2o_u_b = load['Meth_H2O_flow'][2]
        self.stats_names = list(STATS_NAMES)
