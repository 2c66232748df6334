"""TEST INFRASTRUCTURE ONLY (see oracle/__init__ docstring): per-node CPU restatement of the reference's
tree search, used to check mcts_q_amd.treesearch.LockstepMCTS.  Only tests/ may import it.

Follows /root/reference/src/mctsq_config_mcts_per.py: predict :220-258, _select :260-310, _expand :312-356,
_evaluate :358-370, _backpropagate :372-381, MCTSNode :468-553.  The reference's own MCTSQ class cannot be
imported in this container (it imports gymnasium and tensorboard), so this
restatement is pinned by the source text, not by running the reference: "parity unpinned" for the tree
logic.  The network enters through `q_fn(price, process, gas) -> Q [B, A]` (torch CPU tensors), the
environment through the `_per` protocol env.step([action, state_c])."""
import math

import numpy as np
import torch
import torch.nn.functional as F

PROC = ("T_CAT", "H2_in_MolarFlow", "CH4_syn_MolarFlow", "H2_res_MolarFlow", "H2O_DE_MassFlow", "Elec_Heating")


class Node:
    def __init__(self, state_c, parent=None, action=None, done=False, depth=0, maximum_depth=42):
        self.state_c, self.parent, self.action, self.done = state_c, parent, action, done
        self.depth, self.maximum_depth = depth, maximum_depth
        self.children, self.visits, self.total_value = [], 0, 0.0

    def is_terminal(self):                                           # :508-513
        return self.done or self.depth >= self.maximum_depth

    def legal(self):                                                 # :524-546
        m = self.state_c['Meth_State']
        if m in (0, 1):
            return [0, 1, 2]
        if m in (2, 3, 4):
            return [0, 1, 3, 4]
        return list(range(5))

    def fully_expanded(self):                                        # :515-522
        return set(c.action for c in self.children) == set(self.legal())


def get_state(state_c):                                              # :383-410
    s = state_c['obs_norm']
    price = np.array(s["Elec_Price"])[..., np.newaxis]
    process = np.stack([s[k] for k in PROC], axis=-1)
    gas = np.stack([s["Gas_Price"], s["EUA_Price"]], axis=-1)
    return (torch.FloatTensor(price).unsqueeze(0), torch.FloatTensor(process).unsqueeze(0),
            torch.FloatTensor(gas).unsqueeze(0))


class RefSearch:
    """One tree, searched node by node as the reference does."""

    def __init__(self, q_fn, env, iterations, maximum_depth, c_base, c_init, rng, push=None):
        self.q_fn, self.env, self.rng, self.push = q_fn, env, rng, push
        self.iterations, self.maximum_depth, self.c_base, self.c_init = iterations, maximum_depth, c_base, c_init
        self.root_node, self.tree_remain, self.treedepth = None, False, 0

    def predict(self, state_c, train_eval="train"):                  # :220-258
        self.train_eval, self.treedepth = train_eval, 0
        if self.root_node is not None and self.tree_remain and train_eval == "train":
            root = self.root_node
            stack = [(root, 0)]
            while stack:
                n, d = stack.pop()
                n.depth = d
                stack.extend((c, d + 1) for c in n.children)
        else:
            root = Node(state_c, maximum_depth=self.maximum_depth)
            self.root_node, self.tree_remain = root, True
        for _ in range(self.iterations):
            node = self._select(root)
            if not node.is_terminal():
                node = self._expand(node)
            value = self._evaluate(node.state_c)
            while node is not None:                                  # :378-381
                node.visits += 1
                node.total_value += value
                node = node.parent
        self.root = root
        return max(root.children, key=lambda c: c.visits).action

    def advance(self, action):                                       # learn(), :153-167
        if self.root_node is not None:
            match = [c for c in self.root_node.children if c.action == action]
            if match:
                match[0].parent = None
                self.root_node = match[0]
            else:
                self.root_node = None

    def _select(self, node):                                         # :266-308
        while not node.is_terminal() and node.fully_expanded():
            total_visits = sum(c.visits for c in node.children)
            c_puct = math.log((1 + total_visits + self.c_base) / self.c_base) + self.c_init
            parts = [get_state(c.state_c) for c in node.children]
            with torch.no_grad():
                q = self.q_fn(torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts]),
                              torch.cat([p[2] for p in parts]))
                prior = F.softmax(q, dim=-1)
            for i, c in enumerate(node.children):
                p = prior[i, c.action].item()
                mean_q = c.total_value / c.visits if c.visits > 0 else 0
                c.puct_score = mean_q + c_puct * p * math.sqrt(total_visits) / (1 + c.visits)
            node = max(node.children, key=lambda c: c.puct_score)
        return node

    def _expand(self, node):                                         # :312-356
        if node.depth >= self.maximum_depth:
            return node
        tried = [c.action for c in node.children]
        untried = [a for a in node.legal() if a not in tried]
        action = self.rng.choice(untried)
        next_state, reward, terminated, truncated, info = self.env.step([action, node.state_c])
        state_c = info['state_c']
        if self.train_eval == "train" and self.push is not None:
            self.push(state_c['obs_norm'], action, reward, next_state, terminated)
        self.treedepth = max(self.treedepth, node.depth + 1)
        child = Node(state_c, parent=node, action=action, done=terminated or truncated, depth=node.depth + 1,
                     maximum_depth=self.maximum_depth)
        node.children.append(child)
        return child

    def _evaluate(self, state_c):                                    # :358-370
        with torch.no_grad():
            return self.q_fn(*get_state(state_c)).max().item()
