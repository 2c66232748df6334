"""GPU: mctsq_per_sample_noalloc (the entry point that allocates nothing: its sampling workspace is a fixed array
in the library's device image) against mctsq_per_sample on the same ring and uniforms - indices, physical slots
and IS weights bit for bit - on two calls in a row, the second with a smaller batch; the size limit is refused."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ring(n=50000, cap=65536, seed=3):
    from mcts_q_amd.per import DevicePER
    rng = np.random.RandomState(seed)
    buf = DevicePER(cap)
    rows = np.zeros((n, 91), dtype=np.float32)
    buf.push_rows(rows, np.zeros(n), np.zeros(n), rows, np.zeros(n))
    pr = np.abs(rng.standard_normal(n)).astype(np.float32)
    pr[::9] = 0.0
    buf.update_priorities(np.arange(n), pr)
    return buf, rng


def _noalloc(buf, u, beta=0.4):
    from mcts_q_amd._lib import lib, ptr
    u = torch.as_tensor(u, dtype=torch.float64).cuda().contiguous()
    B = u.numel()
    idx = torch.full((B,), -7, dtype=torch.int64, device="cuda")
    phys = torch.full((B,), -7, dtype=torch.int64, device="cuda")
    w = torch.full((B,), -7.0, dtype=torch.float32, device="cuda")
    wmax = torch.zeros(1, dtype=torch.float64, device="cuda")
    rc = lib.mctsq_per_sample_noalloc(ptr(buf.tree), ptr(buf.prio), buf.capacity, buf.head, buf.size, ptr(u), B,
                                      buf.alpha, beta, buf.mass_capacity, ptr(idx), ptr(phys), ptr(w), ptr(wmax),
                                      torch.cuda.current_stream().cuda_stream)
    return rc, idx, phys, w, wmax


def test_noalloc_sampling_equals_workspace_sampling():
    from mcts_q_amd._lib import lib
    buf, rng = _ring()
    for B in (1024, 100):                                   # second call smaller: stale scratch must not matter
        u = rng.random_sample(B)
        rc, idx, phys, w, wmax = _noalloc(buf, u)
        assert rc == 0, lib.mctsq_last_error()
        idx_r, phys_r, w_r = buf.sample_indices(u)
        torch.cuda.synchronize()
        assert torch.equal(idx, idx_r) and torch.equal(phys, phys_r) and torch.equal(w, w_r)
        assert float(wmax) == float(buf.last_local_wmax) and float(w.max()) == 1.0
    # a wrapped ring (head != 0) through the same entry point
    n = buf.capacity - buf.size + 500
    rows = np.zeros((n, 91), dtype=np.float32)
    buf.push_rows(rows, np.zeros(n), np.zeros(n), rows, np.zeros(n))
    assert buf.head == 500
    u = rng.random_sample(257)
    rc, idx, phys, w, _ = _noalloc(buf, u)
    idx_r, phys_r, w_r = buf.sample_indices(u)
    assert rc == 0 and torch.equal(idx, idx_r) and torch.equal(phys, phys_r) and torch.equal(w, w_r)


def test_noalloc_two_streams_in_sequence():
    """Two calls, one after the other, on two different streams of the device (the scratch is keyed by the stream's
    device, nothing is enqueued by the set-up): both equal the workspace path."""
    from mcts_q_amd._lib import lib
    buf, rng = _ring(n=30000, cap=32768, seed=8)
    torch.cuda.synchronize()
    for B, stream in ((700, torch.cuda.Stream()), (64, torch.cuda.Stream())):
        u = rng.random_sample(B)
        with torch.cuda.stream(stream):
            rc, idx, phys, w, _ = _noalloc(buf, u)
        stream.synchronize()                                  # calls that overlap on one GPU would race (header)
        assert rc == 0, lib.mctsq_last_error()
        idx_r, phys_r, w_r = buf.sample_indices(u)
        torch.cuda.synchronize()
        assert torch.equal(idx, idx_r) and torch.equal(phys, phys_r) and torch.equal(w, w_r)


def test_noalloc_refuses_more_draws_than_its_scratch_holds():
    from mcts_q_amd._lib import lib
    buf, rng = _ring(n=2000, cap=2048)
    limit = lib.mctsq_per_sample_noalloc_max_draws()
    assert limit == 16384
    rc, *_ = _noalloc(buf, rng.random_sample(limit + 1))
    assert rc < 0 and b"at most" in lib.mctsq_last_error()
    rc, idx, _, w, _ = _noalloc(buf, rng.random_sample(limit))       # exactly the limit is served
    assert rc == 0 and int(idx.min()) >= 0 and float(w.max()) == 1.0
