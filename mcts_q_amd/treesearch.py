"""Lock-step MCTS over K independent trees: the host-side caller that turns the reference's per-node
B = 1 network calls into one coalesced leaf batch per search iteration.

Restates, tree for tree, the search of MCTSQ.predict / _select / _expand / _evaluate / _backpropagate and
MCTSNode (/root/reference/src/mctsq_config_mcts_per.py:220-381, 468-553): PUCT selection with the
exploration constant C(s) = log((1 + N + c_base) / c_base) + c_init, random choice among the untried legal
actions, value = max_a Q(leaf), prior = softmax(Q(child))[child.action], visit / value backup to the root,
most-visited child as the result, legal actions by `Meth_State`, sub-tree reuse between environment steps.
The serial tree logic stays on the host (vectorised over trees with numpy); the only device work is the
leaf batch.

Two facts make one launch per iteration enough without changing any result:
  * the network weights do not change inside predict(), so the quantities the reference recomputes on
    every visit (the prior of a child, the value of a terminal node it lands on again) are constants of
    the node: they are taken from the single evaluation the node gets when it is created;
  * a retained sub-tree is re-evaluated in one batch when predict() starts (the weights moved since).

All tree arithmetic is float64 in the reference's operation order, so with the same Q values the chosen
actions and visit counts are identical to K sequential reference searches (tests/test_treesearch_*.py
check this against oracle/mcts_ref.py).  Children are kept in creation order: `max(children, key=...)`
returns the first maximum in that order (mcts_per.py:308, 553).
"""
from __future__ import annotations

import ctypes
import math
import os
import random
import time

import numpy as np

from .leafbatch import N_PRICE, ROW, obs_to_row

N_ACT = 5
# MCTSNode.get_legal_actions (mcts_per.py:524-546) as bit masks over actions 0..4
_LEGAL = {0: 0b00111, 1: 0b00111, 2: 0b11011, 3: 0b11011, 4: 0b11011}
_ALL = 0b11111
_ACTIONS_OF = [[a for a in range(N_ACT) if m >> a & 1] for m in range(32)]
_NBITS = np.array([len(a) for a in _ACTIONS_OF], dtype=np.int64)
_KTH = np.array([(a + [0] * N_ACT)[:N_ACT] for a in _ACTIONS_OF], dtype=np.int64)      # k-th set bit of a mask
_LEGAL_TAB = np.array([_LEGAL.get(m, _ALL) for m in range(8)], dtype=np.int64)


def legal_mask(meth_state) -> int:
    try:
        return _LEGAL.get(int(meth_state), _ALL)
    except (TypeError, ValueError):
        return _ALL


_TREE_LIB = None


def _mix64(z):
    z = z + np.uint64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def lockstep_uniform(seed: int, iteration: int, k, stream: int):
    """mctsq_lockstep_uniform (csrc_host/lockstep.c) for an array of tree indices: the counter-based uniform in
    [0, 1) of (seed, iteration, tree, stream).  Both drivers (numpy and C) draw from it, so they build the same trees."""
    with np.errstate(over="ignore"):
        it = np.array([iteration], dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + np.uint64(stream)
        x = _mix64(np.array([seed], dtype=np.uint64) ^ _mix64(it) ^
                   _mix64(np.asarray(k, dtype=np.uint64) + np.uint64(0x632BE59BD9B4E019)))
    return (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


class Forest(ctypes.Structure):
    """mctsq_forest (include/mctsq_tree.h)."""
    _fields_ = ([(k, ctypes.c_int64) for k in ("K", "cap", "n", "D", "row_w")] +
                [(k, ctypes.c_void_p) for k in ("root", "parent", "action", "visits", "depth", "legal", "tried", "nchild",
                                                "child", "total", "vmax", "prior_a", "svec", "reward", "done", "rows")])


class LeafEval(ctypes.Structure):
    """mctsq_leaf_eval (include/mctsq_tree.h): what mctsq_lockstep_search needs to launch and await leaf batches."""
    _fields_ = ([(k, ctypes.c_void_p) for k in ("forward", "event_record", "event_sync", "memcpy_async", "net", "params",
                                                "prep")] + [("stream", ctypes.c_void_p * 4)] +
                [(k, ctypes.c_int64) for k in ("halves", "cap", "ws_bytes", "n_act")] +
                [(k, ctypes.c_void_p * 4) for k in ("stage", "dev_in", "q", "vmax", "prior", "argmax", "ws", "event")] +
                [("stream_write32", ctypes.c_void_p), ("done_flag", ctypes.c_void_p * 4)])


class HostStagedEvaluator:
    """Adapter giving a plain evaluate(rows) -> (vmax, prior) function the staged interface the C driver feeds
    (stage_ptr / launch / wait / vmax_ptr / prior_ptr, `halves` independent staging areas of `cap` slots each).
    mcts.DeviceLeafEvaluator is the device implementation of the same interface."""

    def __init__(self, evaluate, cap: int, row_w: int = ROW, n_act: int = N_ACT, halves: int = 2):
        self.evaluate, self.cap, self.row_w, self.n_act, self.halves = evaluate, int(cap), int(row_w), int(n_act), halves
        self._stage = [np.zeros(self.cap * self.row_w, dtype=np.float32) for _ in range(halves)]
        self._vmax = [np.zeros(self.cap, dtype=np.float32) for _ in range(halves)]
        self._prior = [np.zeros((self.cap, self.n_act), dtype=np.float32) for _ in range(halves)]

    def __call__(self, rows):
        return self.evaluate(rows)

    def begin(self):
        pass

    def stage_ptr(self, h):
        return self._stage[h].ctypes.data

    def vmax_ptr(self, h):
        return self._vmax[h].ctypes.data

    def prior_ptr(self, h):
        return self._prior[h].ctypes.data

    def launch(self, h, n):
        cap, w = self.cap, self.row_w
        st = self._stage[h]
        P, G = N_PRICE, 6
        L6 = w - P - G
        rows = np.concatenate([st[:cap * P].reshape(cap, P)[:n], st[cap * P:cap * (P + L6)].reshape(cap, L6)[:n],
                               st[cap * (P + L6):].reshape(cap, G)[:n]], axis=1)
        vmax, prior = self.evaluate(rows)
        self._vmax[h][:n] = np.asarray(vmax, dtype=np.float32)
        self._prior[h][:n] = np.asarray(prior, dtype=np.float32)

    def wait(self, h):
        pass


def _tree_lib():
    """libmctsq_tree.so (mcts_q_amd/csrc_host/tree_ops.c): selection and backup over all trees in C, spread over
    the host cores.  Raises if the library was not built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _TREE_LIB
    if _TREE_LIB is None:
        import ctypes
        path = os.environ.get("MCTSQ_TREE_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmctsq_tree.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with make -C mcts_q_amd/csrc")
        L = ctypes.CDLL(path)
        P, I64, F64 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_double
        L.mctsq_tree_select.argtypes = [I64, P, I64, F64, F64, P, P, P, P, P, P, P, P, P]
        L.mctsq_tree_select.restype = None
        L.mctsq_tree_backup.argtypes = [I64, P, P, P, P, P]
        L.mctsq_tree_backup.restype = None
        L.mctsq_tree_retain.argtypes = [I64, I64] + [P] * 15 + [I64, P, I64, P, P]
        L.mctsq_tree_retain.restype = I64
        U64 = ctypes.c_uint64
        L.mctsq_lockstep_expand.argtypes = [ctypes.POINTER(Forest), P, I64, F64, F64, U64, U64, I64, I64, F64, P, P, P, I64, P]
        L.mctsq_lockstep_expand.restype = I64
        L.mctsq_lockstep_backup.argtypes = [ctypes.POINTER(Forest), I64, I64, P, P, I64, P, P, I64]
        L.mctsq_lockstep_backup.restype = None
        L.mctsq_lockstep_search.argtypes = [ctypes.POINTER(Forest), P, I64, F64, F64, U64, U64, I64, F64,
                                            ctypes.POINTER(LeafEval), P, P, P, P]
        L.mctsq_lockstep_search.restype = I64
        L.mctsq_lockstep_uniform.argtypes = [U64, U64, U64, U64]
        L.mctsq_lockstep_uniform.restype = F64
        _TREE_LIB = L
    return _TREE_LIB


class LockstepMCTS:
    """K trees advanced together.

    evaluate(rows) -> (vmax, prior): rows fp32 [n, 91] (leafbatch.obs_to_row layout) -> fp32 [n] max_a Q and
    fp32 [n, A] softmax(Q); ONE device launch per call (DeviceQNet.forward).
    env_step(i, action, state_c) -> (next_obs, reward, terminated, truncated, info): the `_per` environment
    protocol, a pure function of (action, state_c) with info['state_c'] the successor (ptg_gym_env_per.py:422-431).
    push(state, action, reward, next_state, terminated): replay sink for transitions met while expanding in
    training mode (mcts_per.py:336-340), or None.

    Batched-environment mode (batch_env given): the per-tree Python calls above are
    the host bottleneck, so an environment that can step many states at once plugs in through
        batch_env.state_dim                          -> D, length of one state vector (float64)
        batch_env.observe(states [n, D])             -> rows fp32 [n, 91], meth_state int [n]
        batch_env.step_batch(states [n, D], actions) -> next_states [n, D], rows fp32 [n, 91], reward [n],
                                                        terminated [n], truncated [n], meth_state int [n]
    and the driver keeps state vectors instead of state_c dicts; predict() then takes states [K, D], the
    untried action is drawn from ONE numpy generator for all trees (same distribution as random.choice, not
    the same stream), and transitions go to push_rows(rows, actions, rewards, next_rows, dones) in one call
    per iteration.
    """

    def __init__(self, n_trees: int, evaluate, iterations: int, maximum_depth: int, c_base: float, c_init: float,
                 rngs=None, batch_env=None, seed: int = 0, native: bool = True):
        self.K = int(n_trees)
        # native: selection / backup loops in C (libmctsq_tree.so); False keeps them in numpy (same arithmetic, used
        # by the tests to check the C path)
        self.native = bool(native)
        self._lib = _tree_lib() if self.native else None
        self.evaluate = evaluate
        self.iterations = int(iterations)
        self.maximum_depth = int(maximum_depth)
        self.c_base, self.c_init = c_base, c_init
        # one random stream per tree; None = the global `random` module, as the reference uses (:328)
        self.rngs = list(rngs) if rngs is not None else [random] * self.K
        assert len(self.rngs) == self.K
        self.batch_env = batch_env
        self.D = int(batch_env.state_dim) if batch_env is not None else 0
        self.seed, self.iter_count = int(seed) & (2 ** 64 - 1), 0
        self.prof = None
        self._compact = False       # True right after _retain: every stored node hangs under a current root
        # C driver: the whole iteration (selection, action choice, environment step, node creation / backup) in one
        # call per block of trees, when the environment is the C one (ptgenv.PtGBatchEnv exposes its handle as c_env)
        self.fast = bool(self.native and batch_env is not None and getattr(batch_env, "c_env", None))
        self.root = np.full(self.K, -1, dtype=np.int64)
        self.treedepth = np.zeros(self.K, dtype=np.int64)
        self._alloc(0)
        self.n = 0
        self._logtab = []
        if self.fast:
            if int(batch_env.row_width) != ROW:
                raise ValueError(f"the leaf batch layout needs {ROW}-scalar observations, the environment has {batch_env.row_width}")
            if not (hasattr(evaluate, "launch") or hasattr(evaluate, "c_struct")):
                self.evaluate = HostStagedEvaluator(evaluate, -(-self.K // 2) if self.K >= 64 else self.K, ROW, N_ACT,
                                                    2 if self.K >= 64 else 1)

    # ---- storage ------------------------------------------------------------------------------
    def _alloc(self, cap):
        self.cap = cap
        self.parent = np.full(cap, -1, dtype=np.int64)
        self.action = np.full(cap, -1, dtype=np.int64)
        self.visits = np.zeros(cap, dtype=np.int64)
        self.total = np.zeros(cap, dtype=np.float64)
        self.depth = np.zeros(cap, dtype=np.int64)
        self.done = np.zeros(cap, dtype=bool)
        self.legal = np.zeros(cap, dtype=np.int64)
        self.tried = np.zeros(cap, dtype=np.int64)
        self.nchild = np.zeros(cap, dtype=np.int64)
        self.child = np.full((cap, N_ACT), -1, dtype=np.int64)      # creation order
        self.vmax = np.zeros(cap, dtype=np.float64)
        self.prior_a = np.zeros(cap, dtype=np.float64)
        self.rows = self._alloc_rows(cap)
        self.svec = np.zeros((cap, self.D), dtype=np.float64)       # batched-environment mode: state vectors
        self.reward = np.zeros(cap, dtype=np.float64)               # reward of the transition that created the node
        self._forest = None
        self.state_c = [None] * cap
        self.extra = [None] * cap           # (meth_state_tr, el_price_tr, pot_reward_tr) of mcts_per.py:351-353

    def _alloc_rows(self, cap):
        """Observation rows; with the C driver on a GPU host they live in pinned memory, so the replay push of a whole
        search's expansions is one direct DMA."""
        if getattr(self, "fast", False) and cap:
            try:
                import torch
                if torch.cuda.is_available():
                    self._rows_pin = torch.zeros((cap, ROW), dtype=torch.float32, pin_memory=True)
                    return self._rows_pin.numpy()
            except ImportError:
                pass
        return np.zeros((cap, ROW), dtype=np.float32)

    def _grow(self, need):
        if need <= self.cap:
            return
        old = self.__dict__.copy()
        cap = max(need, 2 * self.cap, 1024)
        self._alloc(cap)
        n = old["n"] if "n" in old else 0
        for k in ("parent", "action", "visits", "total", "depth", "done", "legal", "tried", "nchild", "child",
                  "vmax", "prior_a", "rows", "svec", "reward"):
            getattr(self, k)[:n] = old[k][:n]
        self.state_c[:n] = old["state_c"][:n]
        self.extra[:n] = old["extra"][:n]

    def _new_nodes(self, state_cs, parents, actions, dones, depths, extras=None):
        """Append nodes (MCTSNode.__init__, :472-506); returns their ids."""
        m = len(state_cs)
        self._grow(self.n + m)
        ids = np.arange(self.n, self.n + m)
        self.n += m
        self.parent[ids], self.action[ids] = parents, actions
        self.visits[ids], self.total[ids] = 0, 0.0
        self.depth[ids], self.done[ids] = depths, dones
        self.tried[ids], self.nchild[ids] = 0, 0
        self.child[ids] = -1
        for j, i in enumerate(ids):
            sc = state_cs[j]
            self.state_c[i] = sc
            self.extra[i] = extras[j] if extras is not None else None
            self.legal[i] = legal_mask(sc['Meth_State'])
            self.rows[i] = obs_to_row(sc['obs_norm'])
        return ids

    def _new_nodes_batch(self, svec, rows, meth, parents, actions, dones, depths):
        """Batched-environment form of _new_nodes: no per-node Python."""
        m = len(svec)
        self._grow(self.n + m)
        ids = np.arange(self.n, self.n + m)
        self.n += m
        self.parent[ids], self.action[ids] = parents, actions
        self.visits[ids], self.total[ids] = 0, 0.0
        self.depth[ids], self.done[ids] = depths, dones
        self.tried[ids], self.nchild[ids] = 0, 0
        self.child[ids] = -1
        self.svec[ids], self.rows[ids] = svec, rows
        meth = np.asarray(meth, dtype=np.int64)
        self.legal[ids] = np.where((meth >= 0) & (meth < 8), _LEGAL_TAB[np.clip(meth, 0, 7)], _ALL)
        return ids

    def _c_puct(self, tv):
        """math.log((1 + N + c_base) / c_base) + c_init per total visit count (:273), tabulated."""
        top = int(tv.max()) if len(tv) else 0
        while len(self._logtab) <= top:
            n = len(self._logtab)
            self._logtab.append(math.log((1 + n + self.c_base) / self.c_base) + self.c_init)
        return np.asarray(self._logtab, dtype=np.float64)[tv]

    def _evaluate_nodes(self, ids):
        """ids: an index array, or a slice (all nodes: no gather copy of the rows)."""
        m = (ids.stop - ids.start) if isinstance(ids, slice) else len(ids)
        if m == 0:
            return
        vmax, prior = self.evaluate(self.rows[ids])
        self.vmax[ids] = np.asarray(vmax, dtype=np.float32).astype(np.float64)          # q.max().item() (:370)
        prior = np.asarray(prior, dtype=np.float32)
        act = np.where(self.action[ids] >= 0, self.action[ids], 0)
        self.prior_a[ids] = prior[np.arange(m), act].astype(np.float64)                 # prior[idx, action].item() (:296)

    # ---- sub-tree reuse (mcts_per.py:153-167, 236-247) -----------------------------------------
    def _retain(self, keep_roots):
        """Compact the arrays to the sub-trees under keep_roots[i] (>= 0), re-rooted at depth 0."""
        K = self.K
        self._compact = True
        if self.n == 0 or not (keep_roots >= 0).any():
            self.n = 0
            return np.full(K, -1, dtype=np.int64)
        if self.native and self.batch_env is not None:            # no per-node Python objects to carry along
            keep_roots = np.ascontiguousarray(keep_roots, dtype=np.int64)
            new_roots = np.empty(K, dtype=np.int64)
            rel, newid = np.empty(self.n, dtype=np.int64), np.empty(self.n, dtype=np.int64)
            self.n = int(self._lib.mctsq_tree_retain(
                K, self.n, keep_roots.ctypes.data, new_roots.ctypes.data, self.parent.ctypes.data,
                self.action.ctypes.data, self.visits.ctypes.data, self.total.ctypes.data, self.depth.ctypes.data,
                self.done.ctypes.data, self.legal.ctypes.data, self.tried.ctypes.data, self.nchild.ctypes.data,
                self.child.ctypes.data, self.vmax.ctypes.data, self.prior_a.ctypes.data, self.rows.ctypes.data,
                self.rows.shape[1], self.svec.ctypes.data, self.svec.shape[1], rel.ctypes.data, newid.ctypes.data))
            return new_roots
        mark = np.zeros(self.n, dtype=bool)
        rel = np.zeros(self.n, dtype=np.int64)
        frontier = keep_roots[keep_roots >= 0]
        mark[frontier] = True
        d = 0
        while len(frontier):
            rel[frontier] = d
            ch = self.child[frontier].ravel()
            frontier = ch[ch >= 0]
            mark[frontier] = True
            d += 1
        old_ids = np.nonzero(mark)[0]
        remap = np.full(self.n + 1, -1, dtype=np.int64)           # remap[-1] stays -1
        remap[old_ids] = np.arange(len(old_ids))
        for k in ("action", "visits", "total", "done", "legal", "tried", "nchild", "vmax", "prior_a", "rows", "svec"):
            arr = getattr(self, k)
            arr[:len(old_ids)] = arr[old_ids]
        self.depth[:len(old_ids)] = rel[old_ids]                 # reset_depths (:239-243)
        self.parent[:len(old_ids)] = remap[self.parent[old_ids]]
        self.child[:len(old_ids)] = remap[self.child[old_ids]]
        if self.batch_env is None:                                # per-environment mode keeps Python objects per node
            self.state_c[:len(old_ids)] = [self.state_c[i] for i in old_ids]
            self.extra[:len(old_ids)] = [self.extra[i] for i in old_ids]
        self.n = len(old_ids)
        new_roots = np.where(keep_roots >= 0, remap[keep_roots], -1)
        self.parent[new_roots[new_roots >= 0]] = -1               # detach from the previous root (:162)
        return new_roots

    def advance(self, actions, keep=None):
        """After the environment took `actions`: the matching child becomes the next root (:154-165);
        trees with keep[i] False (episode end, tree_remain interval, validation) start fresh."""
        actions = np.asarray(actions, dtype=np.int64)
        ok = self.root >= 0
        if keep is not None:
            ok &= np.asarray(keep, dtype=bool)
        ch = self.child[np.where(ok, self.root, 0)]                       # [K, 5]
        hit = (ch >= 0) & (self.action[np.where(ch >= 0, ch, 0)] == actions[:, None]) & ok[:, None]
        nxt = np.where(hit.any(1), ch[np.arange(self.K), np.argmax(hit, axis=1)], -1)
        self.root = self._retain(nxt)

    def reset(self, mask=None):
        """Forget the trees with mask[i] True (all if None)."""
        keep = self.root.copy()
        if mask is None:
            keep[:] = -1
        else:
            keep[np.asarray(mask, dtype=bool)] = -1
        self.root = self._retain(keep)

    def _select_numpy(self, trees, maxd):
        """Selection for all trees level by level (the arithmetic tree_ops.c repeats)."""
        cur = self.root.copy()
        while True:
            terminal = self.done[cur] | (self.depth[cur] >= maxd)
            go = ~terminal & (self.tried[cur] == self.legal[cur])
            if not go.any():
                break
            t = trees[go]
            node = cur[t]
            ch = self.child[node]                                     # [n, 5] creation order
            valid = ch >= 0
            chs = np.where(valid, ch, 0)
            v = np.where(valid, self.visits[chs], 0)
            tv = v.sum(1)
            c_puct = self._c_puct(tv)
            vf = v.astype(np.float64)
            mean = np.where(v > 0, self.total[chs] / np.where(v > 0, vf, 1.0), 0.0)
            explore = c_puct[:, None] * self.prior_a[chs] * np.sqrt(tv.astype(np.float64))[:, None] / (1 + vf)
            score = np.where(valid, mean + explore, -np.inf)
            cur[t] = ch[np.arange(len(t)), np.argmax(score, axis=1)]
        return cur

    # ---- the search ---------------------------------------------------------------------------
    def predict(self, state_cs, env_step=None, train: bool = True, push=None, reuse=None, push_rows=None):
        """One MCTS search per tree from state_cs[i]; returns the most-visited root action of every tree
        (int64 [K]).  reuse[i] True keeps tree i's retained sub-tree (if advance() left one).  In
        batched-environment mode state_cs is the [K, D] state array and env_step is unused."""
        K = self.K
        benv = self.batch_env
        assert len(state_cs) == K
        self.treedepth[:] = 0
        keep = self.root.copy()
        if reuse is None or not train:                 # a retained tree is only used in training mode (:236)
            keep[:] = -1
        else:
            keep[~np.asarray(reuse, dtype=bool)] = -1
        if not (self._compact and np.array_equal(keep, self.root)):    # advance() already left exactly these sub-trees
            self.root = self._retain(keep)
        self._compact = False
        fresh = np.nonzero(self.root < 0)[0]
        if len(fresh) and benv is not None:
            sv = np.asarray(state_cs, dtype=np.float64)[fresh]
            rows, meth = benv.observe(sv)
            self.root[fresh] = self._new_nodes_batch(sv, rows, meth, -1, -1, False, 0)
        elif len(fresh):
            ids = self._new_nodes([state_cs[i] for i in fresh], -1, -1, False, 0)
            self.root[fresh] = ids
        self._evaluate_nodes(slice(0, self.n))         # fresh roots + retained sub-trees under the current weights
        trees = np.arange(K)
        maxd = self.maximum_depth
        if self.fast:
            n0 = self.n
            self._search_fast(maxd)
            if train and push_rows is not None and self.n > n0:       # the expansions of this search, in creation order
                new = slice(n0, self.n)
                push_rows(self.rows[new], self.action[new], self.reward[new], self.rows[new], self.done[new])
            return self._most_visited(trees)
        for _ in range(self.iterations):
            # -- selection (:266-308)
            if self.native:
                cur = np.empty(K, dtype=np.int64)
                self._lib.mctsq_tree_select(K, self.root.ctypes.data, maxd, float(self.c_base), float(self.c_init),
                                            self.done.ctypes.data, self.depth.ctypes.data, self.legal.ctypes.data,
                                            self.tried.ctypes.data, self.child.ctypes.data, self.visits.ctypes.data,
                                            self.total.ctypes.data, self.prior_a.ctypes.data, cur.ctypes.data)
            else:
                cur = self._select_numpy(trees, maxd)
            # -- expansion (:312-356)
            terminal = self.done[cur] | (self.depth[cur] >= maxd)
            exp = trees[~terminal]
            if len(exp) and benv is not None:
                par = cur[exp]
                mask = self.legal[par] & ~self.tried[par] & _ALL
                u = lockstep_uniform(self.seed, self.iter_count, exp, 0)
                k = np.minimum((u * _NBITS[mask]).astype(np.int64), _NBITS[mask] - 1)
                acts = _KTH[mask, k]
                sigma = float(getattr(benv, "noise", 0.0) or 0.0)
                if hasattr(benv, "noise"):                # index noise of mode changes from the same counter-based stream
                    u1, u2 = (lockstep_uniform(self.seed, self.iter_count, exp, q) for q in (1, 2))
                    nz = sigma * np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(6.283185307179586 * u2) if sigma else np.zeros(len(exp))
                    nsv, rows, rew, term, trunc, meth = benv.step_batch(self.svec[par], acts, noise=nz)
                else:
                    nsv, rows, rew, term, trunc, meth = benv.step_batch(self.svec[par], acts)
                if train and push_rows is not None:
                    push_rows(rows, acts, rew, rows, term)                # state == next state here (:337-340)
                ids = self._new_nodes_batch(nsv, rows, meth, par, acts, np.asarray(term, dtype=bool) | np.asarray(trunc, dtype=bool),
                                            self.depth[par] + 1)
                self.child[par, self.nchild[par]] = ids
                self.nchild[par] += 1
                self.tried[par] |= 1 << acts
                self.treedepth[exp] = np.maximum(self.treedepth[exp], self.depth[ids])
                cur[exp] = ids
                self._evaluate_nodes(ids)
            elif len(exp):
                scs, acts, dones, extras = [], [], [], []
                for i in exp:
                    node = cur[i]
                    untried = _ACTIONS_OF[self.legal[node] & ~self.tried[node] & _ALL]
                    a = self.rngs[i].choice(untried)
                    next_obs, reward, terminated, truncated, info = env_step(i, a, self.state_c[node])
                    sc = info['state_c']
                    if train and push is not None:
                        # the reference pushes state == next state's observation here (:337-340)
                        push(sc['obs_norm'], a, reward, next_obs, terminated)
                    scs.append(sc); acts.append(a); dones.append(bool(terminated or truncated))
                    extras.append((info.get('Meth_State'), info.get('el_price_act'), info.get('Pot_Reward')))
                par = cur[exp]
                ids = self._new_nodes(scs, par, np.asarray(acts), np.asarray(dones), self.depth[par] + 1, extras)
                par = cur[exp]                                            # (arrays may have been regrown)
                self.child[par, self.nchild[par]] = ids
                self.nchild[par] += 1
                self.tried[par] |= 1 << np.asarray(acts)
                self.treedepth[exp] = np.maximum(self.treedepth[exp], self.depth[ids])
                cur[exp] = ids
                self._evaluate_nodes(ids)                                  # ONE launch: the iteration's leaves
            # -- evaluation + backup (:358-381)
            if self.native:
                cur = np.ascontiguousarray(cur, dtype=np.int64)
                self._lib.mctsq_tree_backup(K, cur.ctypes.data, self.vmax.ctypes.data, self.parent.ctypes.data,
                                            self.visits.ctypes.data, self.total.ctypes.data)
            else:
                value = self.vmax[cur]
                node = cur.copy()
                live = node >= 0
                while live.any():
                    nl = node[live]
                    self.visits[nl] += 1
                    self.total[nl] += value[live]
                    node[live] = self.parent[nl]
                    live = node >= 0
            self.iter_count += 1
        return self._most_visited(trees)

    def _forest_struct(self):
        f = self._forest
        if f is None:
            f = self._forest = Forest()
            f.K, f.cap, f.D, f.row_w = self.K, self.cap, self.D, self.rows.shape[1]
            for k in ("parent", "action", "visits", "depth", "legal", "tried", "nchild", "child", "total", "vmax",
                      "prior_a", "svec", "reward", "done", "rows"):
                setattr(f, k, getattr(self, k).ctypes.data)
        f.n, f.root = self.n, self.root.ctypes.data            # predict() replaces the root array
        return f

    def _search_fast(self, maxd):
        """The iterations of predict() through the C driver, the trees in `halves` blocks that alternate between the
        host (selection / environment / backup) and the device (leaf batch): while one block's leaves are being
        evaluated the other block is advanced.  Per tree the order select -> expand -> evaluate -> backup and every
        random draw are those of the numpy path."""
        K, ev, L = self.K, self.evaluate, self._lib
        self._grow(self.n + K * self.iterations)
        f = self._forest_struct()
        fp = ctypes.byref(f)
        nh = ev.halves if K >= 2 * 32 else 1
        bounds = [(h * K // nh, (h + 1) * K // nh) for h in range(nh)]
        if max(hi - lo for lo, hi in bounds) > ev.cap:
            raise ValueError("leaf staging area smaller than a block of trees")
        cur = np.empty(K, dtype=np.int64)
        slot_all = np.empty(K, dtype=np.int64)
        slots = [slot_all[lo:hi] for lo, hi in bounds]
        pend = [None] * nh
        env, sigma = self.batch_env.c_env, float(self.batch_env.noise)
        cb, ci, A = float(self.c_base), float(self.c_init), N_ACT
        curp, tdp = cur.ctypes.data, self.treedepth.ctypes.data
        ev.begin()

        if hasattr(ev, "c_struct"):          # device evaluator: the whole loop in ONE C call (no Python per iteration)
            le = ev.c_struct()
            if nh != le.halves or max(hi - lo for lo, hi in bounds) > le.cap:
                raise ValueError("leaf evaluator blocks do not match the driver's")
            parr = (ctypes.c_double * 4)(0, 0, 0, 0) if self.prof is not None else None
            rc = L.mctsq_lockstep_search(fp, env, maxd, cb, ci, self.seed, self.iter_count, self.iterations, sigma,
                                         ctypes.byref(le), curp, slot_all.ctypes.data, tdp, parr)
            if rc < 0:
                # the iterations that completed stay (their nodes are linked and counted); the failing one was taken
                # back by the C driver, so the forest is consistent and the next predict() can go on from it
                self.n = int(f.n)
                self.iter_count += self.iterations
                if hasattr(ev, "abort"):
                    ev.abort()                                   # streams drained, completion words reset
                raise (IndexError(f"environment step failed for {-1 - rc} states (clock outside the price tables, or "
                                  "staging / node capacity exceeded)") if rc > -1000 else
                       RuntimeError(f"leaf batch launch failed with code {-1000 - rc}"))
            if parr is not None:
                for q, k in enumerate(("wait", "backup", "expand", "launch")):
                    self.prof[k] += parr[q]
            self.iter_count += self.iterations
            self.n = int(f.n)
            return
        prof = self.prof                     # optional wall-clock split {wait, backup, expand, launch} (tools/tree_bench.py)
        clock = time.perf_counter

        def finish(h):
            n_new = pend[h]
            t0 = clock() if prof is not None else 0.0
            if n_new:
                ev.wait(h)
            t1 = clock() if prof is not None else 0.0
            lo, hi = bounds[h]
            L.mctsq_lockstep_backup(fp, lo, hi, curp, slots[h].ctypes.data, n_new, ev.vmax_ptr(h), ev.prior_ptr(h), A)
            pend[h] = None
            if prof is not None:
                prof["wait"] += t1 - t0
                prof["backup"] += clock() - t1

        for _ in range(self.iterations):
            for h, (lo, hi) in enumerate(bounds):
                if pend[h] is not None:
                    finish(h)
                t0 = clock() if prof is not None else 0.0
                n_new = L.mctsq_lockstep_expand(fp, env, maxd, cb, ci, self.seed, self.iter_count, lo, hi, sigma, curp,
                                                slots[h].ctypes.data, ev.stage_ptr(h), ev.cap, tdp)
                if n_new < 0:
                    for hh in range(nh):                         # nothing stays in flight; what completed stays in the forest
                        if pend[hh] is not None:
                            finish(hh)
                    self.n = int(f.n)
                    raise IndexError(f"environment step failed for {-1 - n_new} states (clock outside the price tables, "
                                     "or staging / node capacity exceeded)")
                t1 = clock() if prof is not None else 0.0
                if n_new:
                    ev.launch(h, n_new)
                pend[h] = n_new
                if prof is not None:
                    prof["expand"] += t1 - t0
                    prof["launch"] += clock() - t1
            self.iter_count += 1
        for h in range(nh):
            if pend[h] is not None:
                finish(h)
        self.n = int(f.n)

    def _most_visited(self, trees):
        # -- most visited child, first maximum in creation order (:256, 553)
        if (self.nchild[self.root] == 0).any():
            raise ValueError("most_visited_child of a root without children (terminal root or iterations == 0)")
        ch = self.child[self.root]
        v = np.where(ch >= 0, self.visits[np.where(ch >= 0, ch, 0)], -1)
        best = ch[trees, np.argmax(v, axis=1)]
        return self.action[best]
