"""Vectorised PtG environment (mcts_q_amd/ptgenv.py -> libmctsq_tree.so, host code: no GPU needed) against the scalar
restatement of the reference environment (oracle/ptgenv_ref.py) - SURVEY.md 8f-2.

PARITY UNPINNED against the reference itself: its environment imports gymnasium (absent here) and it ships no
environment fixtures.  What is checked: on sequential episodes driven by several action policies (thousands of
(state, action) pairs, every cell of the action x mode table, every branch of the load-change rules), the batched step
returns the oracle's successor state (indices, modes, data-set ids: exact), the oracle's observation (fp32 rows:
exact), termination flag, and the reward to 1e-12 relative (interval means come from extended-precision prefix sums
instead of np.average's pairwise sums).  On synthetic data everywhere; on the reference's own CSV / YAML files when
/root/reference is present (dev container)."""
import os

import numpy as np
import pytest

from mcts_q_amd.leafbatch import obs_to_row
from mcts_q_amd.ptgenv import DATASETS, PtGBatchEnv
from mcts_q_amd.synthetic_env import synthetic_ptg_kwargs
from oracle.ptgenv_ref import RefPtGEnv

REF = "/root/reference"


def _policy(kind, rng, m, t):
    if kind == "uniform":
        return int(rng.randint(5))
    if kind == "sticky":                                  # long stretches in one mode: reaches the ends of the data sets
        return int(rng.randint(5)) if rng.random_sample() < 0.08 else [0, 1, 2, 3, 4][m]
    if kind == "cycle":                                   # startup -> partial <-> full with short and long dwell times
        if m in (0, 1):
            return 2
        if m == 2:
            return 2
        return int(rng.choice([3, 4], p=[0.5, 0.5])) if rng.random_sample() < 0.35 else m
    raise AssertionError(kind)


def _rollout(kw, kind, seed, steps, d0=0, coverage=None):
    ref, env = RefPtGEnv(kw), PtGBatchEnv(kw, seed=0)
    rng = np.random.RandomState(seed)
    obs, info = ref.reset(d0)
    st = env.reset([0])
    st[0, 10] = d0
    sc = info["state_c"]
    rows, meth = env.observe(st)
    assert np.array_equal(rows[0], obs_to_row(obs)) and meth[0] == sc["Meth_State"]
    assert st[0, 0] == sc["i"] and st[0, 4] == sc["Meth_T_cat"]
    worst = 0.0
    for t in range(steps):
        a = _policy(kind, rng, sc["Meth_State"], t)
        nz = float(rng.normal(0, kw["noise"]))
        if coverage is not None:
            coverage.add((a, sc["Meth_State"], sc["part_op"], sc["full_op"]))
        obs, rew, term, _, info = ref.step(a, sc, nz)
        nst, rows, r, tm, tr, meth = env.step_batch(st, [a], noise=[nz])
        sc = info["state_c"]
        head = [sc["i"], sc["j"], sc["k"], sc["Meth_State"], sc["Meth_T_cat"], sc["standby_op"], sc["startup_op"],
                sc["part_op"], sc["full_op"], info["Meth_Hot_Cold"], d0]
        assert np.array_equal(nst[0, :11], np.asarray(head, dtype=np.float64)), (t, a, nst[0, :11], head)
        assert np.array_equal(rows[0], obs_to_row(obs)), (t, a)
        assert bool(tm[0]) == bool(term) and not tr[0] and meth[0] == sc["Meth_State"]
        worst = max(worst, abs(r[0] - rew) / max(abs(rew), 1e-3))
        assert abs(nst[0, 11] - info["cum_reward"]) <= 1e-10 * max(1.0, abs(info["cum_reward"]))
        inf = env.info(nst)
        for k in ("el_price_act", "gas_price_act", "eua_price_act", "Pot_Reward", "Part_Full"):
            assert inf[k][0] == info[k], k
        st = nst
        if term:
            break
    assert worst <= 1e-12, worst
    return t + 1


@pytest.mark.parametrize("kind,seed,scale", [("uniform", 1, 0.2), ("sticky", 2, 0.2), ("cycle", 3, 0.2), ("sticky", 4, 1.0),
                                             ("cycle", 5, 1.0), ("uniform", 6, 1.0)])
def test_sequential_episodes_match_the_oracle(kind, seed, scale):
    kw = synthetic_ptg_kwargs(seed, scale=scale)
    n = _rollout(kw, kind, seed, 560, d0=int(seed % 2) * 4)
    assert n >= 500


def test_every_cell_of_the_mode_table_and_every_load_change_branch_is_exercised():
    kw = synthetic_ptg_kwargs(11, scale=0.2)
    cov = set()
    for kind, seed in (("uniform", 21), ("sticky", 22), ("cycle", 23), ("cycle", 24), ("sticky", 25)):
        _rollout(kw, kind, seed, 570, coverage=cov)
    cells = {(a, m) for a, m, _, _ in cov}
    assert cells == {(a, m) for a in range(5) for m in range(5)}
    ids = {n: i for i, n in enumerate(DATASETS)}
    full_ops = {fl for a, m, _, fl in cov if (a, m) == (3, 4)}           # _partial is entered from these full-load sets
    part_ops = {pt for a, m, pt, _ in cov if (a, m) == (4, 3)}
    assert {ids["op2_start_f"], ids["op3_p_f"]} <= full_ops and len(full_ops) >= 4
    assert {ids["op1_start_p"], ids["op8_f_p"]} <= part_ops and len(part_ops) >= 4


def test_scenarios_and_terminal_step():
    for scenario in (1, 3):
        kw = synthetic_ptg_kwargs(7, scale=0.2, env_values=dict(scenario=scenario))
        _rollout(kw, "cycle", 30 + scenario, 120)
    kw = synthetic_ptg_kwargs(8, scale=0.2, env_values=dict(eps_len_d=1), train_days=9)
    assert kw["eps_sim_steps"] == 144
    assert _rollout(kw, "cycle", 9, 400) == 144 - 5                          # terminated at k == eps_sim_steps - 6


def test_batched_step_is_pure_and_equals_single_steps():
    kw = synthetic_ptg_kwargs(3, scale=0.2)
    env = PtGBatchEnv(kw, seed=5)
    rng = np.random.RandomState(0)
    st = env.reset(np.arange(700))
    for _ in range(40):                                                   # spread the states over modes and clocks
        st = env.step_batch(st, rng.randint(0, 5, len(st)))[0]
    act, nz = rng.randint(0, 5, len(st)), rng.normal(0, 10, len(st))
    big = env.step_batch(st, act, noise=nz)                               # >= 128 states: the OpenMP loop
    again = env.step_batch(st.copy(), act, noise=nz)
    for x, y in zip(big, again):
        assert np.array_equal(x, y)
    perm = rng.permutation(len(st))
    shuf = env.step_batch(st[perm], act[perm], noise=nz[perm])
    for x, y in zip(big, shuf):
        assert np.array_equal(x[perm], y)
    for s in range(0, 700, 37):
        one = env.step_batch(st[s:s + 1], act[s:s + 1], noise=nz[s:s + 1])
        for x, y in zip(big, one):
            assert np.array_equal(x[s:s + 1], y)
    rows, meth = env.observe(big[0])
    assert np.array_equal(rows, big[1]) and np.array_equal(meth, big[5])
    assert len(np.unique(big[5])) == 5


def test_state_c_interchange_and_errors():
    kw = synthetic_ptg_kwargs(4, scale=0.2)
    env, ref = PtGBatchEnv(kw), RefPtGEnv(kw)
    obs, info = ref.reset(0)
    sc = info["state_c"]
    for a in (2, 2, 2, 3):
        obs, _, _, _, info = ref.step(a, sc, 0.0)
        sc = info["state_c"]
    named = dict(sc, standby_op=DATASETS[sc["standby_op"]], startup_op=DATASETS[sc["startup_op"]],
                 part_op=DATASETS[sc["part_op"]], full_op=DATASETS[sc["full_op"]])
    st = env.state_from_state_c(named, hot_cold=info["Meth_Hot_Cold"], cum_reward=info["cum_reward"])
    rows, _ = env.observe(st[None])
    assert np.array_equal(rows[0], obs_to_row(obs))
    with pytest.raises(IndexError):
        env.step_batch(st[None], [7])
    late = st.copy(); late[2] = 10 ** 6
    with pytest.raises(IndexError):
        env.observe(late[None])
    bad = dict(kw); bad["seq_step"] = 40                                   # 9 * 40 > 299: window longer than a step
    with pytest.raises(ValueError):
        PtGBatchEnv(bad)
    bad = dict(kw); bad["op3_p_f"] = kw["op3_p_f"][:100]
    with pytest.raises(ValueError):
        PtGBatchEnv(bad)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "data", "OP2")), reason="reference data files not present")
def test_on_the_reference_data_files():
    """Dev container only: the reference's own YAML + CSV files through load_data / Preprocessing (8f-4), then the same
    oracle comparison on the real tables; the look-ahead arrays are re-derived by direct slicing."""
    import contextlib, io
    from mcts_q_amd.envconfig import EnvConfiguration
    from mcts_q_amd.ptgdata import Preprocessing, TrainConfiguration, load_data
    cfg = os.path.join(REF, "config")
    ec = EnvConfiguration(os.path.join(cfg, "config_env.yaml"))
    tc = TrainConfiguration(os.path.join(cfg, "config_train.yaml"), os.path.join(cfg, "config_mctsq.yaml"),
                            os.path.join(cfg, "config_env.yaml"))
    tc.path, tc.seed_train = REF, tc.r_seed_train[0]
    with contextlib.redirect_stdout(io.StringIO()):
        prices, ops = load_data(ec, tc)
        pre = Preprocessing(prices, ops, ec, tc)
    assert ec.train_len_d % ec.eps_len_d == 0 and pre.n_eps == ec.train_len_d // ec.eps_len_d
    assert ops["cooldown"].shape == (45001, 7) and ops["cooldown"][1, 1] == 596.6
    el = prices["el_price_train"]
    assert pre.e_r_b_train.shape == (3, 25, len(el) - 25)
    for i in (0, 12, 24):
        assert np.array_equal(pre.e_r_b_train[0, i], el[i:len(el) - 25 + i])
        assert np.array_equal(pre.e_r_b_train[1, i], pre.dict_pot_r_b["pot_rew_train"][i:len(el) - 25 + i])
    assert np.array_equal(pre.g_e_train[1, 1], prices["eua_price_train"][1:])
    assert sorted(pre.eps_ind[:pre.n_eps]) == list(range(pre.n_eps))
    kw = pre.dict_env_kwargs("train")
    assert kw["eps_sim_steps"] == ec.eps_len_d * 144
    for kind, seed in (("uniform", 41), ("sticky", 42), ("cycle", 43)):
        _rollout(kw, kind, seed, 1200, d0=int(pre.eps_ind[seed % 7]) * ec.eps_len_d)
    kwt = pre.dict_env_kwargs("test")
    assert kwt["eps_ind"] is None and kwt["state_change_penalty"] == 0.0
    _rollout(kwt, "cycle", 44, 600)
