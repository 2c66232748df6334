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
