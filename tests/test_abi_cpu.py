"""The C-ABI library loads without a GPU and exports every entry point include/mctsq_hip.h declares; the
ctypes table (mcts_q_amd/_lib.py) covers the same set.  No compute calls here."""
import ctypes
import os
import re

from mcts_q_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "mctsq_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mctsq_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 25
    dll = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(dll, n)]
    assert not missing, missing
    assert dll.mctsq_abi_version() == 5


def test_ctypes_table_matches_header():
    assert sorted(_lib.SIGNATURES) == _declared()


def test_host_side_queries_need_no_gpu():
    from mcts_q_amd import NetSpec
    spec = NetSpec.from_cfg(dict(price="gru", process="gru", gas="gru", act="relu", flavor="per"))
    d = spec.c_desc()
    lib = _lib.lib
    assert lib.mctsq_q_prepared_bytes(ctypes.byref(d)) == (3 * 4 * 3 * 2 * 3 + 8 * 6 * 3 + 8 * 4 * 3 + 4 * 3) * 64 * 16   # W_hh x 3 + head layers
    assert lib.mctsq_q_forward_workspace_bytes(ctypes.byref(d), 4096) > 4096 * 192 * 4
    assert lib.mctsq_per_tree_words(1_000_000) > 1_000_000
    assert lib.mctsq_q_forward_workspace_bytes(ctypes.byref(d), -1) < 0
    assert b"negative" in lib.mctsq_last_error()


def test_host_tree_helper_exports_its_header():
    """libmctsq_tree.so (host-only C helper of the lock-step search) exports what include/mctsq_tree.h declares."""
    text = open(os.path.join(ROOT, "include", "mctsq_tree.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(mctsq_tree_[a-z0-9_]+)\s*\(", text)))
    assert names == ["mctsq_tree_backup", "mctsq_tree_retain", "mctsq_tree_select", "mctsq_tree_threads"]
    dll = ctypes.CDLL(os.path.join(ROOT, "mcts_q_amd", "lib", "libmctsq_tree.so"))
    assert all(hasattr(dll, n) for n in names)
    env = sorted(set(re.findall(r"\b(mctsq_ptg_[a-z0-9_]+)\s*\(", text)))
    assert env == ["mctsq_ptg_create", "mctsq_ptg_destroy", "mctsq_ptg_info", "mctsq_ptg_observe", "mctsq_ptg_reset",
                   "mctsq_ptg_state_dim", "mctsq_ptg_step", "mctsq_ptg_step_one"]
    assert all(hasattr(dll, n) for n in env)
    drv = sorted(set(re.findall(r"\b(mctsq_lockstep_[a-z0-9_]+)\s*\(", text)))
    assert drv == ["mctsq_lockstep_backup", "mctsq_lockstep_expand", "mctsq_lockstep_search", "mctsq_lockstep_uniform"]
    assert all(hasattr(dll, n) for n in drv)


def test_encoder_layout_query_needs_no_gpu():
    """mctsq_encoder_param_layout: the stand-alone encoders' tensors in named_parameters() order."""
    import numpy as np
    lib = _lib.lib
    for kind, S, F, count, total_min in ((1, 25, 1, 8, 64 * 3 + 64 * 64 * 3), (2, 10, 6, 8, 192 * 6 + 192 * 64),
                                         (3, 10, 6, 29, 500 * 64 + 2 * 2 * 2048 * 64), (4, 3, 2, 4, 64 * 6 + 64 * 64)):
        d = _lib.EncoderDesc(kind, S, F, 0)
        offs, sizes, total = np.zeros(32, np.int64), np.zeros(32, np.int64), ctypes.c_int64(0)
        n = lib.mctsq_encoder_param_layout(ctypes.byref(d), offs.ctypes.data, sizes.ctypes.data, 32, ctypes.addressof(total))
        assert n == count and offs[0] == 0 and total.value >= total_min
        assert all(o % 16 == 0 for o in offs[:n]) and all(offs[i] + sizes[i] <= offs[i + 1] for i in range(n - 1))
        assert offs[n - 1] + sizes[n - 1] <= total.value
        assert lib.mctsq_encoder_workspace_bytes(ctypes.byref(d), 100) >= 112 * 192 * 4
    bad = _lib.EncoderDesc(4, 5, 2, 0)                       # mlp needs seq_len 3
    assert lib.mctsq_encoder_workspace_bytes(ctypes.byref(bad), 4) < 0
