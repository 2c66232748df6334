"""Observation dict <-> canonical 91-scalar state row, and leaf-batch assembly.

Mirror of MCTSQ._get_state (/root/reference/src/mctsq_config_mcts_per.py:383-410) and of the batch
assembly in DQNModel.update (src/mctsq_config_dqn_per.py:415-452): price window first, then the six
process traces stacked feature-last in the reference's fixed key order, then gas / EUA.  Row layout:
price[25] | process[10][6] (time-major) | gas_eua[3][2]; the kernels read it as three tensors
[B,25,1], [B,10,6], [B,3,2].
"""
from __future__ import annotations

import numpy as np
import torch

PROCESS_KEYS = ("T_CAT", "H2_in_MolarFlow", "CH4_syn_MolarFlow", "H2_res_MolarFlow", "H2O_DE_MassFlow",
                "Elec_Heating")
N_PRICE, N_PROC, N_GAS = 25, 10, 3
ROW = N_PRICE + N_PROC * 6 + N_GAS * 2       # 91


def obs_to_row(obs: dict) -> np.ndarray:
    """fp32 row of one normalised observation dict (fp64 arrays are cast like torch.FloatTensor does)."""
    proc = np.stack([np.asarray(obs[k]) for k in PROCESS_KEYS], axis=-1)
    ge = np.stack([np.asarray(obs["Gas_Price"]), np.asarray(obs["EUA_Price"])], axis=-1)
    return np.concatenate([np.asarray(obs["Elec_Price"]).ravel(), proc.ravel(), ge.ravel()]).astype(np.float32)


def row_to_obs(row) -> dict:
    row = np.asarray(row)[:ROW]
    n_proc = (len(row) - N_PRICE - N_GAS * 2) // 6
    proc = row[N_PRICE:N_PRICE + n_proc * 6].reshape(n_proc, 6)
    ge = row[N_PRICE + n_proc * 6:].reshape(-1, 2)
    obs = {"Elec_Price": row[:N_PRICE].copy()}
    for i, k in enumerate(PROCESS_KEYS):
        obs[k] = proc[:, i].copy()
    obs["Gas_Price"], obs["EUA_Price"] = ge[:, 0].copy(), ge[:, 1].copy()
    return obs


def split_rows(rows: torch.Tensor):
    """[K,91+] device/host tensor -> (price [K,25,1], process [K,10,6], gas [K,3,2]) contiguous."""
    K = rows.shape[0]
    return (rows[:, :25].reshape(K, 25, 1).contiguous(), rows[:, 25:85].reshape(K, 10, 6).contiguous(),
            rows[:, 85:91].reshape(K, 3, 2).contiguous())


def pack_leaves(observations, device="cuda:0", pinned: torch.Tensor | None = None):
    """Gather the leaf states of many trees into ONE coalesced batch: a single pinned host staging
    buffer, a single H2D copy, three views.  `observations` is a sequence of obs dicts."""
    K = len(observations)
    host = pinned if pinned is not None and pinned.shape[0] >= K else torch.empty(K, ROW, dtype=torch.float32)
    buf = host.numpy()
    for i, o in enumerate(observations):
        buf[i] = obs_to_row(o)
    return split_rows(host[:K].to(device, non_blocking=True))
