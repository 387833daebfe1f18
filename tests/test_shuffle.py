"""go2sim_shuffle_gather (include/go2sim.h): the head of PPO.update — one permutation of the rollout for all epochs (rsl_rl/rsl_rl/storage/rollout_storage.py:150)
and the storage tensors gathered into mini-batch order — as one library call.  Here: the oracle and the host build of the HIP library; GPU twin in
tests/test_gpu_shuffle.py.  The permutation is a keyed bijection computed per row (6 Feistel rounds + cycle walking, include/go2sim_shuffle.h), so it is tested
as what PPO needs of torch.randperm: a bijection for every key, a new one every call, and statistically uniform (every source row equally likely at every
position, adjacent positions independent)."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import load_emu, load_oracle
from go2_rl_gym_amd._abi import Go2GatherJob

WIDTHS = (45, 263, 12, 1, 1, 1, 1, 12, 12)          # the nine tensors of PPO._KEYS


def gather(lib, rows, device="cpu", indices=None, key=None, widths=WIDTHS, seed=0):
    g = torch.Generator().manual_seed(seed)
    src = [torch.randn(rows, w, generator=g).to(device) for w in widths]
    dst = [torch.full((rows, w), float("nan"), device=device) for w in widths]
    jobs = (Go2GatherJob * len(widths))(*[Go2GatherJob(s.data_ptr(), d.data_ptr(), w, 0) for s, d, w in zip(src, dst, widths)])
    clear = torch.ones(3, device=device)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream) if device != "cpu" else None
    rc = lib.go2sim_shuffle_gather(jobs, len(widths), rows, C.c_void_p(indices.data_ptr()) if indices is not None else None,
                                   C.c_void_p(key.data_ptr()) if key is not None else None, C.c_void_p(clear.data_ptr()), 2, stream)
    assert rc == 0, lib.go2sim_last_error().decode()
    return src, dst, clear


def check_explicit_and_keyed(lib, device="cpu", rows=1000):
    idx = torch.randperm(rows, generator=torch.Generator().manual_seed(1)).to(device)
    src, dst, clear = gather(lib, rows, device, indices=idx)
    for s, d in zip(src, dst):
        assert torch.equal(d, s[idx])                      # torch.index_select's result, bit for bit
    assert clear.tolist() == [0.0, 0.0, 1.0]
    key = torch.tensor([777, 5, 0, 0], dtype=torch.int32, device=device)
    perms = []
    for call in range(3):
        src, dst, _ = gather(lib, rows, device, key=key)
        if device != "cpu":
            torch.cuda.synchronize()
        want = torch.tensor([lib.go2sim_shuffle_index(i, rows, 777, 5 + call) for i in range(rows)], device=device)
        assert sorted(want.tolist()) == list(range(rows))          # a bijection
        for s, d in zip(src, dst):
            assert torch.equal(d, s[want])
        assert key.tolist() == [777, 5 + call + 1, 0, 0]           # the counter advanced, the ticket is back at zero
        perms.append(want)
    assert not torch.equal(perms[0], perms[1]) and not torch.equal(perms[1], perms[2])


@pytest.mark.parametrize("which", ["oracle", "emu"])
def test_shuffle_gather_explicit_and_keyed(which):
    check_explicit_and_keyed(load_oracle() if which == "oracle" else load_emu())


def test_shuffle_refuses_bad_arguments():
    lib = load_oracle()
    t = torch.zeros(8)
    j = (Go2GatherJob * 1)(Go2GatherJob(t.data_ptr(), t.data_ptr(), 2, 0))
    assert lib.go2sim_shuffle_gather(j, 1, 4, None, None, None, 0, None) < 0           # neither indices nor a key
    assert lib.go2sim_shuffle_gather(j, 0, 4, None, C.c_void_p(t.data_ptr()), None, 0, None) < 0
    assert lib.go2sim_shuffle_gather(None, 1, 4, None, C.c_void_p(t.data_ptr()), None, 0, None) < 0
    j[0].row_floats = 0
    assert lib.go2sim_shuffle_gather(j, 1, 4, None, C.c_void_p(t.data_ptr()), None, 0, None) < 0


def test_keyed_permutation_is_statistically_uniform():
    """over 4000 keys on n = 96 rows (a non-power-of-two, like 98304): (a) the source row at a fixed position, (b) the position of a fixed row and (c) the
    difference of two adjacent positions' source rows are uniform by chi-square at the 1e-4 level; and at the update's real size the counts of a fixed position
    spread over the range"""
    lib = load_oracle()
    n, K = 96, 4000
    P = np.array([[lib.go2sim_shuffle_index(i, n, 1000 + 7 * k, k // 3) for i in range(n)] for k in range(K)])
    assert all(sorted(p.tolist()) == list(range(n)) for p in P[:50])
    from scipy.stats import chisquare
    for pos in (0, 1, 47, 95):
        assert chisquare(np.bincount(P[:, pos], minlength=n)).pvalue > 1e-4, pos                       # (a)
    inv = np.argsort(P, axis=1)
    for row in (0, 50, 95):
        assert chisquare(np.bincount(inv[:, row], minlength=n)).pvalue > 1e-4, row                     # (b)
    d = (P[:, 10] - P[:, 11]) % n                                                                    # (c): 0 never occurs (bijection); 1..n-1 equally likely
    assert (d != 0).all() and chisquare(np.bincount(d, minlength=n)[1:]).pvalue > 1e-4
    big = np.array([lib.go2sim_shuffle_index(12345, 98304, s, 3) for s in range(400)])
    assert big.min() < 98304 * 0.05 and big.max() > 98304 * 0.95 and abs(big.mean() / 98304 - 0.5) < 0.06


def keyed_permutation(lib, n, seed, counter):
    """pi as go2sim_shuffle_gather applies it: one library call on a 1-float-per-row tensor (row indices are exact in fp32 up to 2^24)"""
    src = torch.arange(n, dtype=torch.float32)
    dst = torch.empty(n)
    key = torch.tensor([seed, counter, 0, 0], dtype=torch.int32)
    job = (Go2GatherJob * 1)(Go2GatherJob(src.data_ptr(), dst.data_ptr(), 1, 0))
    assert lib.go2sim_shuffle_gather(job, 1, n, None, C.c_void_p(key.data_ptr()), None, 0, None) == 0
    return dst.numpy().astype(np.int64)


def test_minibatch_composition_at_the_update_size():
    """What PPO needs of the permutation beyond uniform positions: WHICH rows share a mini-batch.  At the update's size (4096 envs x 24 steps,
    4 mini-batches) the number of one env's 24 rows that land in mini-batch 0 is hypergeometric (population 98304, 24 marked, 24576 drawn: mean 6, variance 4.4996) — the
    law of a uniformly random permutation (torch.randperm) — checked over 200 draws x 128 envs within 3 sigma of the sampling error; likewise for one STEP's 4096 rows
    (mean 1024, variance 767.9); and consecutive outputs are uncorrelated (|lag-1 correlation| < 3 / sqrt(n))."""
    lib = load_oracle()
    N, T, nmb, draws = 4096, 24, 4, 200
    n, mb = N * T, N * T // nmb
    envs = np.arange(0, N, 32)
    cnt_env, cnt_step, lag = [], [], []
    for d in range(draws):
        p = keyed_permutation(lib, n, 4242 + 13 * (d % 7), d)          # seven seeds, the counter advancing as it does from update to update
        if d < 3:
            assert np.array_equal(np.sort(p), np.arange(n))
        first = p[:mb]                                                 # storage rows (t * N + env) of mini-batch 0
        cnt_env.append(np.bincount(first % N, minlength=N)[envs])
        cnt_step.append(np.bincount(first // N, minlength=T))
        x = p.astype(np.float64)
        lag.append(np.corrcoef(x[:-1], x[1:])[0, 1])
    def check(c, K, what):          # hypergeometric(population n, K marked, mb drawn)
        c = np.concatenate(c).astype(np.float64)
        mean, var = mb * K / n, mb * (K / n) * (1 - K / n) * (n - mb) / (n - 1)
        m = len(c)
        assert abs(c.mean() - mean) < 3 * np.sqrt(var / m), (what, c.mean(), mean)
        # sample variance of m draws: standard error ~ var * sqrt(2 / (m - 1)) (+ the hypergeometric's small excess kurtosis)
        assert abs(c.var(ddof=1) - var) < 4 * var * np.sqrt(2.0 / (m - 1)), (what, c.var(ddof=1), var)
    check(cnt_env, T, "rows of one env in mini-batch 0")
    check(cnt_step, N, "rows of one step in mini-batch 0")
    assert max(abs(v) for v in lag) < 3 / np.sqrt(n) * 1.5, max(abs(v) for v in lag)          # (200 draws: the largest of them, 1.5 x the single-draw 3-sigma)


def cts_indices(lib, nmb, nt, ns, key, device="cpu", mapped=True):
    g = torch.Generator().manual_seed(nt + ns)
    m = (torch.randperm(nt + ns, generator=g) * 3 + 1).to(device) if mapped else None          # an injective map into a larger range
    out = torch.full((nmb * (nt // nmb + ns // nmb),), -1, dtype=torch.int64, device=device)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream) if device != "cpu" else None
    rc = lib.go2sim_cts_minibatch_indices(C.c_void_p(out.data_ptr()), nmb, nt, ns, C.c_void_p(m.data_ptr()) if m is not None else None, C.c_void_p(key.data_ptr()), stream)
    assert rc == 0, lib.go2sim_last_error().decode()
    return out, m


def check_cts_indices(lib, device="cpu", nmb=4, nt=1800, ns=600):
    """go2sim_cts_minibatch_indices against its definition (rollout_storage_cts.py:152-160 with keyed permutations): mini-batch i = [teacher chunk i | student chunk i],
    every sample once, the two populations shuffled independently, a new draw per call"""
    key = torch.tensor([99, 7, 0, 0], dtype=torch.int32, device=device)
    tb, sb = nt // nmb, ns // nmb
    seen = []
    for call in range(3):
        out, m = cts_indices(lib, nmb, nt, ns, key, device)
        if device != "cpu":
            torch.cuda.synchronize()
        o, mm = out.cpu().numpy(), m.cpu().numpy()
        want = np.empty_like(o)
        for i in range(nmb):
            for j in range(tb + sb):
                k = lib.go2sim_shuffle_index(i * tb + j, nt, 99, 7 + call) if j < tb else nt + lib.go2sim_shuffle_index(i * sb + j - tb, ns, 99 ^ 0x9E3779B9, 7 + call)
                want[i * (tb + sb) + j] = mm[k]
        np.testing.assert_array_equal(o, want)
        inv = {int(v): k for k, v in enumerate(mm)}
        ks = np.array([inv[int(v)] for v in o]).reshape(nmb, tb + sb)
        assert (ks[:, :tb] < nt).all() and (ks[:, tb:] >= nt).all() and len(set(ks.reshape(-1).tolist())) == nmb * (tb + sb)
        assert key.tolist() == [99, 7 + call + 1, 0, 0]
        seen.append(o)
    assert not np.array_equal(seen[0], seen[1]) and not np.array_equal(seen[1], seen[2])
    out, _ = cts_indices(lib, 3, 100, 31, torch.tensor([5, 0, 0, 0], dtype=torch.int32, device=device), device, mapped=False)          # ragged: 100 // 3, 31 // 3 (one teacher, one student sample left out)
    o = out.cpu().numpy().reshape(3, 33 + 10)
    assert (o[:, :33] < 100).all() and (o[:, 33:] >= 100).all() and len(set(o.reshape(-1).tolist())) == 129


@pytest.mark.parametrize("which", ["oracle", "emu"])
def test_cts_minibatch_indices(which):
    lib = load_oracle() if which == "oracle" else load_emu()
    check_cts_indices(lib)
    t = torch.zeros(8, dtype=torch.int64)
    assert lib.go2sim_cts_minibatch_indices(C.c_void_p(t.data_ptr()), 4, 3, 8, None, C.c_void_p(t.data_ptr()), None) < 0          # fewer teacher samples than mini-batches
    assert lib.go2sim_cts_minibatch_indices(None, 1, 4, 4, None, C.c_void_p(t.data_ptr()), None) < 0


def test_gather_into_a_column_block_and_long_clear_lists():
    """ABI 8: dst_pitch (gathered rows land in a column block of a wider matrix: the [latent | obs] inputs of CTS) and nclear > 256"""
    for lib in (load_oracle(), load_emu()):
        rows, w, L = 300, 45, 8
        g = torch.Generator().manual_seed(3)
        src, wide, clear = torch.randn(rows, w, generator=g), torch.full((rows, L + w), 5.0), torch.ones(700)
        idx = torch.randperm(rows, generator=g)
        job = (Go2GatherJob * 1)(Go2GatherJob(src.data_ptr(), wide.data_ptr() + 4 * L, w, L + w))
        assert lib.go2sim_shuffle_gather(job, 1, rows, C.c_void_p(idx.data_ptr()), None, C.c_void_p(clear.data_ptr()), 650, None) == 0
        assert torch.equal(wide[:, L:], src[idx]) and (wide[:, :L] == 5.0).all()
        assert (clear[:650] == 0).all() and (clear[650:] == 1).all()
        job[0].dst_pitch = w - 1
        assert lib.go2sim_shuffle_gather(job, 1, rows, C.c_void_p(idx.data_ptr()), None, None, 0, None) < 0          # a pitch shorter than the row
