"""go2sim_shuffle_gather on a real MI355X (tests/test_shuffle.py is the CPU twin): explicit indices == torch.index_select bit for bit, the keyed permutation equals
its host statement go2sim_shuffle_index for every row, the device-side counter advances once per launch (also when the launch is replayed from a HIP graph) at the
update's size.  Run with -m gpu."""
import ctypes as C

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
