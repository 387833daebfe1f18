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
