"""GPU parity: device PER (through the C ABI) vs the reference's golden vectors and the oracle.
Bar: sampled indices and written-back priorities bit-exact; IS weights bit-exact vs the
fixed-point oracle and within one fp32 ulp of the reference's."""
import os

import numpy as np
import pytest
import torch

from oracle import per_numpy

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TAGS = ["small", "wrap", "big", "partial"]


def _rows(n, base=0):
    r = np.zeros((n, 91), dtype=np.float32)
    r[:, 0] = np.arange(base, base + n)           # payload = push ordinal, like the fixture
    return r


def _build(tag, z, chunk=97):
    from mcts_q_amd.per import DevicePER
    cap, n_push = int(z[f"{tag}/capacity"]), int(z[f"{tag}/n_push"])
    buf = DevicePER(cap)
    for s in range(0, n_push, chunk):             # ragged push batches, crossing the wrap point
        k = min(chunk, n_push - s)
        rows = _rows(k, s)
        buf.push_rows(rows, np.arange(s, s + k) % 5, np.arange(s, s + k, dtype=np.float32), rows + 1,
                      np.zeros(k))
    buf.update_priorities(z[f"{tag}/upd_idx"], z[f"{tag}/upd_prio"])
    return buf


@pytest.mark.parametrize("tag", TAGS)
def test_sample_matches_reference_golden(tag):
    z = np.load(os.path.join(GOLDEN, "per_cases.npz"))
    buf = _build(tag, z)
    torch.cuda.synchronize()
    assert len(buf) == len(z[f"{tag}/prio"])
    assert np.array_equal(buf.logical_priorities(), z[f"{tag}/prio"])        # priorities bit-exact
    idx, phys, w = buf.sample_indices(z[f"{tag}/u"])
    assert np.array_equal(idx.cpu().numpy(), z[f"{tag}/idx"])                 # indices bit-exact
    assert np.abs(w.cpu().numpy() - z[f"{tag}/w"]).max() <= 1.2e-7
    _, w_or, _ = per_numpy.fixed_point_sample(z[f"{tag}/prio"], z[f"{tag}/u"], capacity=buf.capacity)
    assert np.array_equal(w.cpu().numpy(), w_or)                              # weights bit-exact vs oracle
    batch = buf.gather(phys)
    assert np.array_equal(batch["sprice"][:, 0, 0].cpu().numpy().astype(np.int64), z[f"{tag}/payload"])
    assert np.array_equal(batch["act"].cpu().numpy(), z[f"{tag}/payload"] % 5)
    assert np.array_equal(batch["rew"].cpu().numpy(), z[f"{tag}/payload"].astype(np.float32))
    assert np.array_equal(batch["nprice"][:, 0, 0].cpu().numpy().astype(np.int64), z[f"{tag}/payload"] + 1)


@pytest.mark.parametrize("tag", TAGS)
def test_push_takes_true_current_max(tag):
    z = np.load(os.path.join(GOLDEN, "per_cases.npz"))
    buf = _build(tag, z)
    prio = buf.logical_priorities()
    m = int(np.argmax(prio))
    buf.update_priorities(np.array([m]), np.array([0.25], dtype=np.float32))
    buf.push_rows(_rows(1, -1), [0], [0.0], _rows(1, -2), [0.0])
    after = buf.logical_priorities()
    assert after[-1] == float(z[f"{tag}/pushed_prio"])
    assert len(buf) == int(z[f"{tag}/len_after"])


def test_masses_and_tree_bit_exact_vs_oracle():
    """Leaf masses (deterministic pow + quantisation) and every tree level equal the oracle's
    integers; size-independent property: root == sum of leaves == prefix sums."""
    from mcts_q_amd.per import DevicePER
    from mcts_q_amd._lib import lib
    rng = np.random.RandomState(5)
    cap = 5000
    buf = DevicePER(cap)
    rows = np.zeros((cap, 91), dtype=np.float32)
    buf.push_rows(rows, np.zeros(cap), np.zeros(cap), rows, np.zeros(cap))
    pr = np.concatenate([np.abs(rng.standard_normal(cap - 6)).astype(np.float32) * 3,
                         np.array([0, 1e-30, 1e-8, 1.0, 255.0, 1e4], dtype=np.float32)])
    buf.update_priorities(np.arange(cap), pr)
    torch.cuda.synchronize()
    shift = lib.mctsq_per_mass_shift(cap)
    assert shift == per_numpy.mass_shift(cap)
    m_or, _ = per_numpy.masses(pr.astype(np.float64), 0.6, shift)
    tree = buf.tree.cpu().numpy().view(np.uint64)
    assert np.array_equal(tree[:cap], m_or)
    lvl1 = np.add.reduceat(np.concatenate([m_or, np.zeros(5056 - cap, dtype=np.uint64)]), np.arange(0, 5056, 64))
    assert np.array_equal(tree[5056:5056 + 79], lvl1)
    assert int(tree[5056 + 128:5056 + 128 + 2].sum()) == int(m_or.sum())


def _check_tree(buf, prio_phys):
    """Every level of the device tree against sums recomputed on the host from the expected leaf masses."""
    from mcts_q_amd._lib import lib
    cap = buf.capacity
    m, _ = per_numpy.masses(np.asarray(prio_phys, dtype=np.float64), buf.alpha, lib.mctsq_per_mass_shift(buf.mass_capacity))
    tree = buf.tree.cpu().numpy().view(np.uint64)
    level, off = m.astype(np.uint64), 0
    while True:
        pad = -(-len(level) // 64) * 64
        assert np.array_equal(tree[off:off + len(level)], level), f"level at word {off}"
        assert not tree[off + len(level):off + pad].any()
        if len(level) <= 64:
            break
        level = np.add.reduceat(np.concatenate([level, np.zeros(pad - len(level), dtype=np.uint64)]), np.arange(0, pad, 64))
        off += pad
    assert off + pad == len(tree)


@pytest.mark.parametrize("cap", [50, 3000, 70_000, 300_000])
def test_tree_levels_after_wrapped_pushes_and_bulk_updates(cap):
    """Push (one workgroup per 4096-slot span rebuilds the nodes the run touches, no per-entry atomics) and the bulk
    priority update (LDS pre-sums for the two top levels): trees of one to four levels, runs that start and end inside
    a span, wrap around the ring, cover it whole; every level equals the sums of the expected leaves, and the sampled
    indices equal the oracle's."""
    from mcts_q_amd.per import DevicePER
    rng = np.random.RandomState(cap)
    buf = DevicePER(cap)
    prio = np.zeros(cap)                                   # by PHYSICAL slot
    head, size = 0, 0
    for K in (max(1, cap // 3), max(1, cap // 2) + 7, min(cap, 4097), 1, cap):
        K = min(K, cap)
        rows = torch.zeros(K, 91, device="cuda")
        buf.push_rows(rows, torch.zeros(K), torch.zeros(K), rows, torch.zeros(K))
        live = prio[(head + np.arange(size)) % cap]
        newp = live.max() if size else 1.0                 # max(priorities, default=1.0) (dqn_per.py:265)
        start = (head + size) % cap
        prio[(start + np.arange(K)) % cap] = newp
        over = max(0, size + K - cap)
        size, head = min(size + K, cap), (head + over) % cap
        assert (buf.head, buf.size) == (head, size)
        _check_tree(buf, prio)
        # bulk update of a random subset with duplicates (> 2048 entries where the ring is big enough)
        n_upd = min(size, 5000) + 3
        li = rng.randint(0, size, n_upd)
        pv = (np.abs(rng.standard_normal(n_upd)) * 2).astype(np.float32)
        pv[::11] = 0.0
        buf.update_priorities(li, pv)
        for i in range(n_upd):                              # sequential semantics: the last duplicate wins
            prio[(head + li[i]) % cap] = float(pv[i])
        _check_tree(buf, prio)
    # consecutive indices (one level-1 node per wave: the wave-sum path) over the whole ring
    pv = np.abs(rng.standard_normal(size)).astype(np.float32)
    buf.update_priorities(np.arange(size), pv)
    prio[(head + np.arange(size)) % cap] = pv.astype(np.float64)
    _check_tree(buf, prio)
    u = rng.random_sample(64)
    idx, _, w = buf.sample_indices(u)
    idx_or, w_or, _ = per_numpy.fixed_point_sample(prio[(head + np.arange(size)) % cap], u, capacity=cap)
    assert np.array_equal(idx.cpu().numpy(), idx_or) and np.array_equal(w.cpu().numpy(), w_or)


def test_large_buffer_statistics_and_roundtrip():
    """BASELINE config 3 size (N = 1M, B = 1024): indices equal the oracle's exactly; empirical
    sampling frequencies follow p_i; update -> sample round trip moves mass as expected."""
    from mcts_q_amd.per import DevicePER
    N, B = 1_000_000, 1024
    rng = np.random.RandomState(3654)
    buf = DevicePER(N)
    chunk = 250_000
    for s in range(0, N, chunk):
        rows = torch.zeros(chunk, 91, device="cuda")
        buf.push_rows(rows, torch.zeros(chunk), torch.zeros(chunk), rows, torch.zeros(chunk))
    pr = np.abs(rng.standard_normal(N)).astype(np.float32)
    pr[: N // 100] = 1.0
    buf.update_priorities(np.arange(N), pr)
    u = np.random.RandomState(3654).random_sample(B)
    idx, phys, w = buf.sample_indices(u)
    idx_or, w_or, _ = per_numpy.fixed_point_sample(pr.astype(np.float64), u, capacity=N)
    assert np.array_equal(idx.cpu().numpy(), idx_or)
    assert np.array_equal(w.cpu().numpy(), w_or)
    idx_ref, w_ref = per_numpy.reference_sample(pr.astype(np.float64), u)
    assert np.array_equal(idx.cpu().numpy(), idx_ref)
    assert np.abs(w.cpu().numpy() - w_ref).max() <= 1.2e-7
    # zero everything but one entry: every draw must return it
    buf.update_priorities(np.arange(N), np.zeros(N, dtype=np.float32))
    buf.update_priorities(np.array([123456, 123456]), np.array([0.0, 2.0], dtype=np.float32))
    idx2, _, w2 = buf.sample_indices(u)
    assert (idx2.cpu().numpy() == 123456).all() and (w2.cpu().numpy() == 1.0).all()


def test_sharded_sampling_kernels_match_single_buffer_oracle():
    """Two rings on one GPU sampled as ONE buffer (concatenation in shard order) through the shard
    entry points: indices, ownership and IS weights equal the single-buffer oracle's bit for bit.
    (The collective plumbing around these calls is covered on CPU by tests/test_parallel_cpu.py.)"""
    from mcts_q_amd.per import DevicePER
    cap, B, world = 3000, 256, 2
    rng = np.random.RandomState(9)
    sizes = [cap, 1700]
    prios = [np.abs(rng.standard_normal(n)).astype(np.float32) for n in sizes]
    prios[0][::13] = 0.0
    shards = []
    for n, pr in zip(sizes, prios):
        s = DevicePER(cap, mass_capacity=world * cap)
        s.push_rows(_rows(n), np.zeros(n), np.zeros(n), _rows(n), np.zeros(n))
        s.update_priorities(np.arange(n), pr)
        shards.append(s)
    u = rng.random_sample(B)
    masses = torch.cat([s.total_mass() for s in shards])
    n_total, mass_total = float(sum(sizes)), float(sum(int(m) for m in masses.tolist()))
    concat = np.concatenate(prios).astype(np.float64)
    idx_or, w_or, _ = per_numpy.fixed_point_sample(concat, u, capacity=world * cap)
    m_or, _ = per_numpy.masses(concat, 0.6, per_numpy.mass_shift(world * cap))
    assert int(masses[0]) == int(m_or[:sizes[0]].sum()) and int(masses[1]) == int(m_or[sizes[0]:].sum())
    parts, gmax = [], torch.zeros(1, dtype=torch.float64, device="cuda")
    for r, s in enumerate(shards):
        targets, mine = s.shard_targets(u, masses, r)
        draw = torch.nonzero(mine).squeeze(1)
        idx, phys, w64, wmax = s.sample_at(targets[draw], n_total, mass_total)
        gmax = torch.maximum(gmax, wmax)
        parts.append((s, draw, idx, w64))
    served = torch.zeros(B, dtype=torch.int32, device="cuda")
    for r, (s, draw, idx, w64) in enumerate(parts):
        w = s.normalize_weights(w64, gmax)
        lo = sum(sizes[:r])
        d = draw.cpu().numpy()
        served[draw] += 1
        assert np.array_equal(idx.cpu().numpy() + lo, idx_or[d])
        assert np.array_equal(w.cpu().numpy(), w_or[d])
    assert (served == 1).all()


@pytest.mark.parametrize("tag", ["wrap", "partial"])
def test_fused_sample_batch_equals_sample_then_gather(tag):
    """mctsq_per_sample_batch (one launch) == mctsq_per_sample followed by mctsq_per_gather, bit for bit,
    twice in a row on the same workspace (the normalisation ticket must return to zero)."""
    z = np.load(os.path.join(GOLDEN, "per_cases.npz"))
    buf = _build(tag, z)
    for rep, n in enumerate((64, 33)):
        u = np.random.RandomState(100 + rep).random_sample(n)
        idx, phys, w = buf.sample_indices(u)
        ref = buf.gather(phys)
        batch, w2, idx2, phys2 = buf.sample_batch(u)
        assert torch.equal(idx, idx2) and torch.equal(phys, phys2) and torch.equal(w, w2)
        for k in ref:
            assert torch.equal(ref[k], batch[k]), k
        assert float(w2.max()) == 1.0


def test_sync_free_shard_protocol_matches_single_buffer_oracle():
    """shard_select + sample_shard (device-side compaction, device-side totals, fused gather): the rows every
    rank serves, in draw order, equal the single-buffer oracle's; padding rows are neutral; a provision that is
    too small is counted, not silently truncated."""
    from mcts_q_amd.per import DevicePER
    cap, B, world = 3000, 256, 2
    rng = np.random.RandomState(19)
    sizes = [cap, 1700]
    prios = [np.abs(rng.standard_normal(n)).astype(np.float32) for n in sizes]
    prios[1][::7] = 0.0
    shards = []
    for n, pr in zip(sizes, prios):
        s = DevicePER(cap, mass_capacity=world * cap)
        s.push_rows(_rows(n, 10000 * len(shards)), np.arange(n) % 5, np.arange(n, dtype=np.float32), _rows(n, 7), np.zeros(n))
        s.update_priorities(np.arange(n), pr)
        shards.append(s)
    u = rng.random_sample(B)
    ms = torch.stack([s.mass_and_size() for s in shards])
    masses, nsz = ms[:, 0].contiguous(), ms[:, 1].contiguous()
    assert nsz.tolist() == sizes
    concat = np.concatenate(prios).astype(np.float64)
    idx_or, w_or, _ = per_numpy.fixed_point_sample(concat, u, capacity=world * cap)
    bcap = 208
    outs, gmax = [], torch.zeros(1, dtype=torch.float64, device="cuda")
    for r, s in enumerate(shards):
        targets, draw, count = s.shard_select(u, masses, r, bcap)
        batch, idx, w64, wmax = s.sample_shard(targets, count, masses, nsz)
        gmax = torch.maximum(gmax, wmax)
        outs.append((draw, count, batch, idx, w64))
    served = np.zeros(B, dtype=np.int64)
    for r, (draw, count, batch, idx, w64) in enumerate(outs):
        lo = sum(sizes[:r])
        mine = np.nonzero((idx_or >= lo) & (idx_or < lo + sizes[r]))[0]
        n = len(mine)
        assert count.tolist() == [n, 0]
        d = draw.cpu().numpy()
        assert np.array_equal(d[:n], mine) and (d[n:] == -1).all()
        served[d[:n]] += 1
        li = idx.cpu().numpy()
        assert np.array_equal(li[:n] + lo, idx_or[mine]) and (li[n:] == -1).all()
        w = shards[r].normalize_weights(w64, gmax).cpu().numpy()
        assert np.array_equal(w[:n], w_or[mine]) and (w[n:] == 0).all()
        assert np.array_equal(batch["sprice"][:n, 0, 0].cpu().numpy().astype(np.int64), li[:n] + 10000 * r)
        assert np.array_equal(batch["act"][:n].cpu().numpy(), li[:n] % 5)
        # padding rows never touch the priorities
        before = shards[r].logical_priorities().copy()
        shards[r].update_priorities(idx[n:], torch.full((bcap - n,), 9.0, device="cuda"))
        assert np.array_equal(shards[r].logical_priorities(), before)
    assert (served == 1).all()
    _, _, count = shards[0].shard_select(u, masses, 0, 32)
    n0 = int(((idx_or >= 0) & (idx_or < sizes[0])).sum())
    assert count.tolist() == [32, n0 - 32]


def test_priority_above_the_mass_clamp():
    """prio^alpha is clamped at 256 (include/mctsq_hip.h): a priority above ~1.03e4 is sampled as if it
    were exactly at the clamp, and the IS weights are those of that law (oracle: the same clamp; reference law on
    the clamped priorities)."""
    from mcts_q_amd.per import DevicePER
    n = 2000
    rng = np.random.RandomState(4)
    pr = np.abs(rng.standard_normal(n)).astype(np.float32)
    pr[[5, 900]] = [3.0e6, 1.0e5]
    buf = DevicePER(n)
    buf.push_rows(_rows(n), np.zeros(n), np.zeros(n), _rows(n), np.zeros(n))
    buf.update_priorities(np.arange(n), pr)
    u = rng.random_sample(512)
    idx, _, w = buf.sample_indices(u)
    idx_or, w_or, _ = per_numpy.fixed_point_sample(pr.astype(np.float64), u, capacity=n)
    assert np.array_equal(idx.cpu().numpy(), idx_or) and np.array_equal(w.cpu().numpy(), w_or)
    clamped = np.minimum(pr.astype(np.float64), 256.0 ** (1 / 0.6))
    idx_ref, w_ref = per_numpy.reference_sample(clamped, u)
    assert np.array_equal(idx.cpu().numpy(), idx_ref)
    assert np.abs(w.cpu().numpy() - w_ref).max() <= 2e-7
    assert (idx_or == 5).sum() > 20 and abs(int((idx_or == 5).sum()) - int((idx_or == 900).sum())) < 40


@pytest.mark.parametrize("B", [1000, 6000])
def test_update_priorities_paths_and_bounds(B):
    """Single-launch path (B <= 4096) and two-launch path: duplicates (last wins), indices at and beyond `size` on
    a ring that is not full (device tensors: skipped; host arrays: IndexError like deque indexing)."""
    from mcts_q_amd.per import DevicePER
    cap, n = 20000, 12000
    rng = np.random.RandomState(B)
    buf = DevicePER(cap)
    buf.push_rows(_rows(n), np.zeros(n), np.zeros(n), _rows(n), np.zeros(n))
    ref = per_numpy.RefPER(cap)
    ref.prio = [1.0] * n
    ref.items = list(range(n))
    idx = rng.randint(0, n, size=B)
    idx[1::5] = idx[0]
    pr = np.abs(rng.standard_normal(B)).astype(np.float32)
    buf.update_priorities(idx, pr)
    ref.update_priorities(idx, pr)
    assert np.array_equal(buf.logical_priorities(), np.asarray(ref.prio))
    m_or, _ = per_numpy.masses(np.asarray(ref.prio), 0.6, per_numpy.mass_shift(cap))
    tree = buf.tree.cpu().numpy().view(np.uint64)
    assert np.array_equal(tree[:n], m_or) and int(buf.total_mass()) == int(m_or.sum())
    # device indices outside [0, size) are ignored: no never-pushed slot becomes sampleable
    bad = torch.tensor([n, cap - 1, -3, n - 1], device="cuda")
    buf.update_priorities(bad, torch.tensor([5.0, 5.0, 5.0, 0.5], device="cuda"))
    ref.update_priorities([n - 1], [np.float32(0.5)])
    assert np.array_equal(buf.logical_priorities(), np.asarray(ref.prio))
    assert not buf.tree.cpu().numpy().view(np.uint64)[n:cap].any()
    with pytest.raises(IndexError):
        buf.update_priorities(np.array([n]), np.array([1.0], dtype=np.float32))
    buf.update_priorities(np.array([-1]), np.array([0.125], dtype=np.float32))    # deque[-1] = newest
    assert buf.logical_priorities()[-1] == 0.125
