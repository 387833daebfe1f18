"""go2sim_shuffle_gather on a real MI355X (tests/test_shuffle.py is the CPU twin): explicit indices == torch.index_select bit for bit, the keyed permutation equals
its host statement go2sim_shuffle_index for every row, the device-side counter advances once per launch (also when the launch is replayed from a HIP graph) at the
update's size.  Run with -m gpu."""
import ctypes as C

import numpy as np

import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import load_hip  # noqa: E402
from test_shuffle import check_explicit_and_keyed, gather  # noqa: E402


@pytest.mark.parametrize("rows", [1000, 33, 4097])
def test_shuffle_gather_on_gpu(rows):
    check_explicit_and_keyed(load_hip(), device="cuda:0", rows=rows)


def test_shuffle_gather_replayed_from_a_graph_at_the_update_size():
    lib = load_hip()
    rows = 98304
    key = torch.tensor([4242, 0, 0, 0], dtype=torch.int32, device="cuda:0")
    src, dst, _ = gather(lib, rows, "cuda:0", key=key)          # eager: counter 0 -> 1
    torch.cuda.synchronize()
    from go2_rl_gym_amd._abi import Go2GatherJob
    jobs = (Go2GatherJob * len(src))(*[Go2GatherJob(s.data_ptr(), d.data_ptr(), s.shape[1], 0) for s, d in zip(src, dst)])
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            assert lib.go2sim_shuffle_gather(jobs, len(src), rows, None, C.c_void_p(key.data_ptr()), None, 0, C.c_void_p(side.cuda_stream)) == 0
    prev = None
    for call in range(3):
        g.replay(); torch.cuda.synchronize()
        assert key.tolist() == [4242, 2 + call, 0, 0]
        sample = torch.arange(0, rows, 997, device="cuda:0")
        want = torch.tensor([lib.go2sim_shuffle_index(int(i), rows, 4242, 1 + call) for i in sample.tolist()], device="cuda:0")
        for s, d in zip(src, dst):
            assert torch.equal(d[sample], s[want])
        # every source row exactly once: the gathered 1-column tensor is a permutation of its source
        assert torch.equal(torch.sort(dst[3].view(-1)).values, torch.sort(src[3].view(-1)).values)
        assert prev is None or not torch.equal(prev, dst[3])
        prev = dst[3].clone()


def test_cts_minibatch_indices_on_gpu():
    """go2sim_cts_minibatch_indices (two keyed permutations -> the CTS update's index list) on the MI355X against its host-side definition, at a toy size and at the
    update's size (3072 teacher + 1024 student envs x 24 steps, 4 mini-batches)"""
    from test_shuffle import check_cts_indices, cts_indices
    lib = load_hip()
    check_cts_indices(lib, "cuda:0")
    key = torch.tensor([31337, 2, 0, 0], dtype=torch.int32, device="cuda:0")
    nt, ns, nmb = 3072 * 24, 1024 * 24, 4
    out, m = cts_indices(lib, nmb, nt, ns, key, "cuda:0")
    torch.cuda.synchronize()
    o = out.cpu().numpy().reshape(nmb, -1)
    inv = torch.empty(int(m.max()) + 1, dtype=torch.int64); inv[m.cpu()] = torch.arange(nt + ns)
    ks = inv[torch.from_numpy(o)].numpy()
    assert (ks[:, :nt // nmb] < nt).all() and (ks[:, nt // nmb:] >= nt).all() and len(np.unique(ks)) == nt + ns and key.tolist() == [31337, 3, 0, 0]


def test_gather_into_a_column_block_on_gpu():
    lib = load_hip()
    rows, w, L = 24576, 263, 32
    g = torch.Generator().manual_seed(3)
    src, wide, clear = torch.randn(rows, w, generator=g).to("cuda:0"), torch.full((rows, L + w), 5.0, device="cuda:0"), torch.ones(700, device="cuda:0")
    idx = torch.randperm(rows, generator=g).to("cuda:0")
    from go2_rl_gym_amd._abi import Go2GatherJob
    job = (Go2GatherJob * 1)(Go2GatherJob(src.data_ptr(), wide.data_ptr() + 4 * L, w, L + w))
    assert lib.go2sim_shuffle_gather(job, 1, rows, C.c_void_p(idx.data_ptr()), None, C.c_void_p(clear.data_ptr()), 650, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    torch.cuda.synchronize()
    assert torch.equal(wide[:, L:], src[idx]) and (wide[:, :L] == 5.0).all() and (clear[:650] == 0).all() and (clear[650:] == 1).all()
