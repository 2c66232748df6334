"""Host logic of the lock-step tree driver (mcts_q_amd/treesearch.py) against the per-node restatement of
the reference's search (oracle/mcts_ref.py), with the SAME CPU network on both sides so that the comparison
isolates the tree logic: chosen actions, visit counts, values and replay pushes must be identical."""
import random

import numpy as np
import pytest
import torch

from mcts_q_amd.leafbatch import split_rows
from mcts_q_amd.treesearch import LockstepMCTS, legal_mask
from oracle import mcts_ref, torch_ref
from mcts_q_amd.synthetic_env import FakePerEnv

CFG = dict(price="gru", process="gru", gas="gru", act="relu", flavor="per")


def _net(seed=0):
    torch.manual_seed(seed)
    return torch_ref.RefQNet(CFG).eval()


def _rowwise(net):
    """torch's CPU kernels are not batch-invariant (blocking changes the fp32 rounding), and the two sides
    batch differently by design; evaluating one row at a time on both sides isolates the tree logic."""
    def q_fn(price, process, gas):
        with torch.no_grad():
            return torch.cat([net(price[i:i + 1], process[i:i + 1], gas[i:i + 1]) for i in range(price.shape[0])])
    return q_fn


def _evaluator(net):
    q_fn = _rowwise(net)

    def ev(rows):
        q = q_fn(*split_rows(torch.from_numpy(np.ascontiguousarray(rows))))
        return q.max(1)[0].numpy(), torch.softmax(q, -1).numpy()
    return ev


def test_legal_masks():
    assert legal_mask(0) == legal_mask(1) == 0b00111
    assert legal_mask(2) == legal_mask(3) == legal_mask(4) == 0b11011
    assert legal_mask(7) == legal_mask("x") == 0b11111


@pytest.mark.parametrize("K,iters,maxd,horizon", [(1, 50, 42, 1000), (5, 30, 3, 1000), (4, 40, 42, 2)])
def test_lockstep_equals_sequential_reference_search(K, iters, maxd, horizon):
    net = _net()
    envs = [FakePerEnv(seed=i, horizon=horizon) for i in range(K)]
    drv_push, ref_push = [[] for _ in range(K)], [[] for _ in range(K)]
    drv = LockstepMCTS(K, _evaluator(net), iters, maxd, c_base=19652, c_init=1.25,
                       rngs=[random.Random(100 + i) for i in range(K)])
    refs = [mcts_ref.RefSearch(_rowwise(net), envs[i], iters, maxd, 19652, 1.25, random.Random(100 + i),
                               push=lambda *a, i=i: ref_push[i].append(a)) for i in range(K)]
    scs = [e.reset()[1]["state_c"] for e in envs]
    ref_scs = list(scs)
    for step in range(4):                                           # four env steps with sub-tree reuse
        cur_i = [None]

        def env_step(i, a, sc):
            cur_i[0] = i
            return envs[i].step([a, sc])

        def push(*a):
            drv_push[cur_i[0]].append(a)

        acts = drv.predict(scs, env_step, train=True, push=push, reuse=[True] * K)
        for i in range(K):
            a_ref = refs[i].predict(ref_scs[i], "train")
            assert acts[i] == a_ref, (step, i)
            root, rr = drv.root[i], refs[i].root
            assert drv.visits[root] == rr.visits and drv.total[root] == rr.total_value
            got = [(drv.action[c], drv.visits[c], drv.total[c]) for c in drv.child[root][:drv.nchild[root]]]
            exp = [(c.action, c.visits, c.total_value) for c in rr.children]
            assert got == exp
            assert drv.treedepth[i] == refs[i].treedepth
            assert len(drv_push[i]) == len(ref_push[i])
            for x, y in zip(drv_push[i], ref_push[i]):
                assert x[1] == y[1] and x[2] == y[2] and x[4] == y[4]
                assert all(np.array_equal(x[0][k], y[0][k]) and np.array_equal(x[3][k], y[3][k]) for k in x[0])
        res = [envs[i].step([int(acts[i]), scs[i]]) for i in range(K)]
        term = np.array([r[2] for r in res])
        # learn(): an episode end resets the environment and drops the tree (mcts_per.py:173-176)
        nxt = [envs[i].reset()[1]["state_c"] if term[i] else res[i][4]["state_c"] for i in range(K)]
        scs, ref_scs = nxt, list(nxt)
        drv.advance(acts, keep=~term)
        for i in range(K):
            refs[i].advance(int(acts[i]))
            if term[i]:
                refs[i].tree_remain = False


def test_eval_mode_starts_fresh_and_does_not_push():
    net = _net(1)
    env = FakePerEnv(3)
    drv = LockstepMCTS(1, _evaluator(net), 20, 42, 19652, 1.25, rngs=[random.Random(5)])
    sc = env.reset()[1]["state_c"]
    pushed = []
    a = drv.predict([sc], lambda i, a, s: env.step([a, s]), train=False, push=lambda *x: pushed.append(x))
    assert pushed == [] and 0 <= a[0] < 5
    assert drv.visits[drv.root[0]] == 20


def test_terminal_root_raises_like_reference():
    net = _net(1)
    env = FakePerEnv(3)
    drv = LockstepMCTS(1, _evaluator(net), 5, 0, 19652, 1.25, rngs=[random.Random(5)])
    with pytest.raises(ValueError):
        drv.predict([env.reset()[1]["state_c"]], lambda i, a, s: env.step([a, s]))


class _PerAdapter:
    """FakeBatchEnv seen through the per-tree `_per` protocol (for the per-node reference search)."""

    def __init__(self, benv):
        self.benv = benv

    def wrap(self, svec, row, meth):
        from mcts_q_amd.leafbatch import row_to_obs
        return {"svec": np.asarray(svec), "obs_norm": row_to_obs(row), "Meth_State": int(meth)}

    def step(self, arg):
        a, sc = arg
        nsv, rows, rew, term, trunc, meth = self.benv.step_batch(sc["svec"][None], np.array([a]))
        nsc = self.wrap(nsv[0], rows[0], meth[0])
        return nsc["obs_norm"], float(rew[0]), bool(term[0]), bool(trunc[0]), {"state_c": nsc}


class _UniformChoice:
    """random.choice driven by the counter-based stream the batched drivers draw from (tree 0, one iteration per
    call: in the test below every iteration expands)."""

    def __init__(self, seed):
        self.seed, self.it = seed, 0

    def choice(self, seq):
        from mcts_q_amd.treesearch import lockstep_uniform
        u = float(lockstep_uniform(self.seed, self.it, [0], 0)[0])
        self.it += 1
        return seq[min(int(u * len(seq)), len(seq) - 1)]


def test_batched_environment_mode_equals_reference_search():
    from mcts_q_amd.synthetic_env import FakeBatchEnv
    net = _net(2)
    benv = FakeBatchEnv(horizon=1000)
    ad = _PerAdapter(benv)
    drv = LockstepMCTS(1, _evaluator(net), 40, 42, 19652, 1.25, batch_env=benv, seed=9)
    ref = mcts_ref.RefSearch(_rowwise(net), ad, 40, 42, 19652, 1.25, _UniformChoice(9))
    st = benv.reset(np.array([5]))
    for step in range(3):
        pushed = []
        a = drv.predict(st, train=True, reuse=[True], push_rows=lambda *x: pushed.append(x))
        rows, meth = benv.observe(st)
        a_ref = ref.predict(ad.wrap(st[0], rows[0], meth[0]), "train")
        assert a[0] == a_ref
        root, rr = drv.root[0], ref.root
        got = [(drv.action[c], drv.visits[c], drv.total[c]) for c in drv.child[root][:drv.nchild[root]]]
        assert got == [(c.action, c.visits, c.total_value) for c in rr.children]
        assert sum(len(p[1]) for p in pushed) == drv.n - (1 if step == 0 else n_kept)
        st = benv.step_batch(st, a)[0]
        drv.advance(a)
        ref.advance(int(a[0]))
        n_kept = drv.n


def test_batched_mode_many_trees_invariants():
    from mcts_q_amd.synthetic_env import FakeBatchEnv
    net = _net(3)
    benv = FakeBatchEnv(horizon=4)
    K, iters = 16, 25
    runs = []
    for _ in range(2):
        drv = LockstepMCTS(K, _evaluator(net), iters, 42, 19652, 1.25, batch_env=benv, seed=4)
        a = drv.predict(benv.reset(np.arange(K)), train=False)
        runs.append((a.copy(), drv.visits[:drv.n].copy()))
        assert (drv.visits[drv.root] == iters).all()
        # every node's visits = 1 (its own evaluation when created or re-selected as terminal) + its children's
        for n in range(drv.n):
            ch = drv.child[n][:drv.nchild[n]]
            term = drv.done[n] or drv.depth[n] >= 42
            if not term and drv.parent[n] >= 0:
                assert drv.visits[n] == 1 + drv.visits[ch].sum()
    assert (runs[0][0] == runs[1][0]).all() and (runs[0][1] == runs[1][1]).all()


@pytest.mark.parametrize("K,iters,horizon", [(1, 30, 3), (37, 20, 4), (300, 6, 6)])
def test_native_select_and_backup_equal_numpy_path(K, iters, horizon):
    """libmctsq_tree.so (csrc_host/tree_ops.c: PUCT selection and value backup over all trees, OpenMP over trees
    from K = 256) against the numpy statements of the same loops: identical trees, bit for bit, over several
    environment steps with sub-tree reuse."""
    from mcts_q_amd.synthetic_env import FakeBatchEnv
    net = _net(5)
    runs = []
    for native in (True, False):
        benv = FakeBatchEnv(horizon=horizon)
        drv = LockstepMCTS(K, _evaluator(net), iters, 42, 19652, 1.25, batch_env=benv, seed=11, native=native)
        states = benv.reset(np.arange(K))
        rec = []
        for step in range(3):
            a = drv.predict(states, train=True, reuse=np.ones(K, dtype=bool))
            rec.append((a.copy(), drv.n, drv.visits[:drv.n].copy(), drv.total[:drv.n].copy(), drv.parent[:drv.n].copy()))
            states, _, _, term, trunc, _ = benv.step_batch(states, a)
            drv.advance(a, keep=~(np.asarray(term, dtype=bool) | np.asarray(trunc, dtype=bool)))
            if (np.asarray(term, dtype=bool) | np.asarray(trunc, dtype=bool)).any():
                break
        runs.append(rec)
    assert len(runs[0]) == len(runs[1])
    for x, y in zip(*runs):
        assert x[1] == y[1]
        for u, v in zip(x, y):
            assert np.array_equal(u, v)


def _hash_eval(rows):
    """Deterministic stand-in for the network, row-wise (so batch composition does not matter)."""
    r = np.asarray(rows, dtype=np.float64)
    q = np.stack([np.sin(r @ (np.arange(1, r.shape[1] + 1) * (0.37 + 0.11 * a)) + a) for a in range(5)], axis=1)
    e = np.exp(q - q.max(1, keepdims=True))
    return q.max(1).astype(np.float32), (e / e.sum(1, keepdims=True)).astype(np.float32)


@pytest.mark.parametrize("K,iters,noise", [(1, 30, 0.0), (37, 20, 10.0), (300, 12, 10.0)])
def test_c_lockstep_driver_equals_numpy_driver_on_the_ptg_environment(K, iters, noise):
    """csrc_host/lockstep.c (one call per iteration and block of trees, blocks alternating as they would around device
    launches) against the numpy driver with the same counter-based draws: identical trees, actions, replay pushes, over
    environment steps with sub-tree reuse; the C uniform equals its numpy statement."""
    from mcts_q_amd.ptgenv import PtGBatchEnv
    from mcts_q_amd.synthetic_env import synthetic_ptg_kwargs
    from mcts_q_amd.treesearch import _tree_lib, lockstep_uniform
    kw = synthetic_ptg_kwargs(2, scale=0.2, env_values=dict(noise=noise))
    L = _tree_lib()
    for it, k, s in ((0, 0, 0), (7, 299, 1), (123456, 5, 2)):
        assert L.mctsq_lockstep_uniform(99, it, k, s) == lockstep_uniform(99, it, [k], s)[0]
    runs = []
    for fast in (True, False):
        env = PtGBatchEnv(kw, seed=1)
        drv = LockstepMCTS(K, _hash_eval, iters, 6, 19652, 1.25, batch_env=env, seed=99)
        assert drv.fast
        drv.fast = fast
        states = env.reset(np.arange(K))
        warm = np.random.RandomState(0)
        for _ in range(5):                                            # leave the initial cooldown state
            states = env.step_batch(states, warm.randint(0, 5, K), noise=np.zeros(K))[0]
        rec = []
        for step in range(3):
            pushed = []
            a = drv.predict(states, train=True, reuse=np.ones(K, dtype=bool), push_rows=lambda *x: pushed.append([np.array(v) for v in x]))
            push = [np.concatenate([p[q] for p in pushed]) for q in range(5)]
            n = drv.n
            rec.append([a.copy(), np.array(n), drv.visits[:n].copy(), drv.total[:n].copy(), drv.parent[:n].copy(),
                        drv.action[:n].copy(), drv.svec[:n].copy(), drv.rows[:n].copy(), drv.treedepth.copy(),
                        drv.done[:n].copy()] + [np.asarray(p, dtype=np.float64) for p in push])
            states = env.step_batch(states, a, noise=np.zeros(K))[0]
            drv.advance(a, keep=np.ones(K, dtype=bool))
        runs.append(rec)
    for x, y in zip(*runs):
        for u, v in zip(x, y):
            assert np.array_equal(u, v)
    assert (runs[0][0][2][:1] >= 1).all() and int(runs[0][2][1]) > K


class _CallbackEvaluator:
    """HostStagedEvaluator whose launch / wait are handed to mctsq_lockstep_search as C function pointers (ctypes
    callbacks standing in for mctsq_q_forward_prepared / hipEventRecord / hipEventSynchronize)."""

    def __init__(self, cap, halves):
        import ctypes
        from mcts_q_amd.treesearch import HostStagedEvaluator, LeafEval
        self.h = HostStagedEvaluator(_hash_eval, cap, halves=halves)
        self.cap, self.halves, self.calls = cap, halves, []
        FWD = ctypes.CFUNCTYPE(ctypes.c_int, *([ctypes.c_void_p] * 6 + [ctypes.c_int64] + [ctypes.c_void_p] * 5 +
                                               [ctypes.c_int64] + [ctypes.c_void_p] * 2))
        EV2 = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)
        EV1 = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p)

        def forward(net, params, prep, price, proc, gas, n, q, vmax, argmax, prior, ws, ws_bytes, stream, mid):
            h = [self.h.stage_ptr(x) for x in range(halves)].index(price)
            assert proc == price + 4 * cap * 25 and gas == price + 4 * cap * 85 and vmax == self.h.vmax_ptr(h)
            self.calls.append((h, n))
            self.h.launch(h, n)
            return 0
        self._keep = (FWD(forward), EV2(lambda e, s: 0), EV1(lambda e: 0))
        le = LeafEval()
        le.forward, le.event_record, le.event_sync = (ctypes.cast(f, ctypes.c_void_p).value for f in self._keep)
        le.halves, le.cap, le.n_act = halves, cap, 5
        for h in range(halves):
            le.stage[h], le.vmax[h], le.prior[h] = self.h.stage_ptr(h), self.h.vmax_ptr(h), self.h.prior_ptr(h)
        self.le = le

    def __call__(self, rows):
        return _hash_eval(rows)

    def begin(self):
        pass

    def c_struct(self):
        return self.le


@pytest.mark.parametrize("K,iters,blocks", [(20, 15, 1), (200, 10, 2), (200, 7, 3), (130, 7, 4)])
def test_whole_search_in_one_c_call_equals_the_per_iteration_driver(K, iters, blocks):
    """mctsq_lockstep_search (the loop over iterations and blocks inside C, leaf batches through function pointers)
    builds the same trees as the per-iteration calls from Python."""
    from mcts_q_amd.ptgenv import PtGBatchEnv
    from mcts_q_amd.synthetic_env import synthetic_ptg_kwargs
    kw = synthetic_ptg_kwargs(6, scale=0.2)
    runs = []
    for one_call in (True, False):
        env = PtGBatchEnv(kw, seed=1)
        halves = blocks
        ev = _CallbackEvaluator(-(-K // halves), halves) if one_call else _hash_eval
        drv = LockstepMCTS(K, ev, iters, 6, 19652, 1.25, batch_env=env, seed=5)
        drv.prof = dict(wait=0.0, backup=0.0, expand=0.0, launch=0.0)
        states = env.reset(np.arange(K))
        rec = []
        for step in range(3):
            a = drv.predict(states, train=True, reuse=np.ones(K, dtype=bool))
            n = drv.n
            rec.append([a.copy(), np.array(n), drv.visits[:n].copy(), drv.total[:n].copy(), drv.parent[:n].copy(),
                        drv.prior_a[:n].copy(), drv.treedepth.copy(), np.array(drv.iter_count)])
            states = env.step_batch(states, a, noise=np.zeros(K))[0]
            drv.advance(a, keep=np.ones(K, dtype=bool))
        runs.append(rec)
        if one_call:
            assert len(ev.calls) == 3 * iters * halves and drv.prof["expand"] > 0
    for x, y in zip(*runs):
        for u, v in zip(x, y):
            assert np.array_equal(u, v)


def _late_states(env, K):
    """States whose clock is a few steps before the end of the price tables: roots observe fine, expansions a few
    levels down step outside the tables (mctsq_ptg_step_one reports them)."""
    st = env.reset(np.arange(K))
    lo, hi = int(st[0, 2]), 10 ** 7
    while hi - lo > 1:                                            # largest clock the environment still observes
        mid = (lo + hi) // 2
        t = st[:1].copy(); t[0, 2] = mid
        try:
            env.observe(t); lo = mid
        except IndexError:
            hi = mid
    return st, lo


@pytest.mark.parametrize("one_call", [True, False])
def test_failed_environment_step_leaves_a_searchable_forest(one_call):
    """A step that fails inside an iteration (clock outside the price tables) raises IndexError and takes the
    iteration's half-made nodes back: the forest the driver keeps is consistent (every stored node is reachable, counted
    and linked exactly once) and the same driver searches healthy states afterwards with the result of a fresh one."""
    from mcts_q_amd.ptgenv import PtGBatchEnv
    from mcts_q_amd.synthetic_env import synthetic_ptg_kwargs
    K, iters = 140, 12
    kw = synthetic_ptg_kwargs(6, scale=0.2)
    env = PtGBatchEnv(kw, seed=1)
    st, last = _late_states(env, K)
    good = np.repeat(st[:1], K, axis=0)                           # K trees from one state (their draws differ)
    bad = good.copy()
    bad[::7, 2] = last - 1                                        # every seventh tree: one step before the end of the tables
    ev = _CallbackEvaluator(-(-K // 2), 2) if one_call else _hash_eval
    drv = LockstepMCTS(K, ev, iters, 6, 19652, 1.25, batch_env=env, seed=5)
    with pytest.raises(IndexError):
        drv.predict(bad, train=False)
    n = drv.n
    assert 0 < n <= K * (iters + 1)
    par, nch = drv.parent[:n], drv.nchild[:n]
    kids = drv.child[:n][drv.child[:n] >= 0]
    assert len(kids) == len(set(kids.tolist())) == int(nch.sum()) and (kids < n).all()
    assert np.array_equal(np.sort(kids), np.nonzero(par >= 0)[0])             # linked children == nodes with a parent
    for c in kids[:200]:
        p = par[c]
        assert c in drv.child[p][:nch[p]] and drv.tried[p] >> drv.action[c] & 1
    assert all(bin(int(drv.tried[i])).count("1") == nch[i] for i in range(n))
    a = drv.predict(good, train=False)                            # the same driver, healthy states
    ref = LockstepMCTS(K, _CallbackEvaluator(-(-K // 2), 2) if one_call else _hash_eval, iters, 6, 19652, 1.25,
                       batch_env=PtGBatchEnv(kw, seed=1), seed=5)
    ref.iter_count = drv.iter_count - iters                       # same counter-based draws
    assert np.array_equal(a, ref.predict(good, train=False))


def test_completion_words_instead_of_events():
    """The pinned-word completion path of mctsq_lockstep_search (stream_write32 / done_flag; the default on the GPU)
    with a host stand-in for hipStreamWriteValue32: same trees as the event path, a NULL done_flag is refused (-3),
    and sequence numbers that wrap around 2^32 still compare as 'written'."""
    import ctypes
    from mcts_q_amd.ptgenv import PtGBatchEnv
    from mcts_q_amd.synthetic_env import synthetic_ptg_kwargs
    from mcts_q_amd.treesearch import _tree_lib
    K, iters = 200, 9
    kw = synthetic_ptg_kwargs(6, scale=0.2)
    WR = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint)

    def write32(stream, ptr, value, flags):
        ctypes.c_uint32.from_address(ptr).value = value
        return 0
    runs = []
    for mode in ("events", "words", "words_wrapping"):
        ev = _CallbackEvaluator(K // 2, 2)
        if mode != "events":
            ev.flags = np.zeros(32, dtype=np.uint32)
            if mode == "words_wrapping":
                ev.flags[:] = 2 ** 32 - 3                         # the sequence numbers pass 2^32 during the search
            ev._wr = WR(write32)
            ev.le.stream_write32 = ctypes.cast(ev._wr, ctypes.c_void_p).value
            for h in range(2):
                ev.le.done_flag[h] = ev.flags.ctypes.data + 64 * h
        drv = LockstepMCTS(K, ev, iters, 6, 19652, 1.25, batch_env=PtGBatchEnv(kw, seed=1), seed=5)
        states = drv.batch_env.reset(np.arange(K))
        a = drv.predict(states, train=False)
        n = drv.n
        runs.append([a.copy(), drv.visits[:n].copy(), drv.total[:n].copy(), drv.parent[:n].copy()])
        if mode == "words":
            assert int(ev.flags[0]) == iters and int(ev.flags[16]) == iters
        if mode == "words_wrapping":
            assert int(ev.flags[0]) == (2 ** 32 - 3 + iters) % 2 ** 32
    for r in runs[1:]:
        for u, v in zip(runs[0], r):
            assert np.array_equal(u, v)
    # a NULL completion word is refused before anything is launched
    ev = _CallbackEvaluator(K // 2, 2)
    ev._wr = WR(write32)
    ev.le.stream_write32 = ctypes.cast(ev._wr, ctypes.c_void_p).value
    drv = LockstepMCTS(K, ev, iters, 6, 19652, 1.25, batch_env=PtGBatchEnv(kw, seed=1), seed=5)
    drv._grow(K * (iters + 2))
    states = drv.batch_env.reset(np.arange(K))
    rows, meth = drv.batch_env.observe(states)
    drv.root[:] = drv._new_nodes_batch(states, rows, meth, -1, -1, False, 0)
    f = drv._forest_struct()
    cur, slots = np.empty(K, dtype=np.int64), np.empty(K, dtype=np.int64)
    rc = _tree_lib().mctsq_lockstep_search(ctypes.byref(f), drv.batch_env.c_env, 6, 19652.0, 1.25, 5, 0, iters, 0.0,
                                           ctypes.byref(ev.le), cur.ctypes.data, slots.ctypes.data,
                                           drv.treedepth.ctypes.data, None)
    assert rc == -3
