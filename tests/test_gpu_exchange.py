"""The in-library exchange of the multi-rank batch rule (pxsom_comm_*, pxsom_batch_train_steps_sharded) on one MI355X:
a 1-rank RCCL communicator owned by libpxsom.  What one GPU can show: RCCL binds (PyTorch's copy, no second
library), the communicator initialises, an all-reduce over one rank leaves the buffer as it was, and a training run
with the exchange enqueued behind every step equals the run without it -- i.e. the per-step call sequence, buffers
and stream are right.  The sum over several ranks is RCCL's; the rule's arithmetic around it is covered by the
2-rank gloo tests (test_distributed_gloo.py, test_pipeline_multirank.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def comm():
    from ark_analysis_amd import som_device
    torch.cuda.set_device(0)
    c = som_device.RankComm(som_device.RankComm.unique_id(), 1, 0)
    yield c
    c.close()


def test_one_rank_allreduce_is_identity(comm):
    t = torch.randn(100 * 23, dtype=torch.float64, device="cuda")
    want = t.clone()
    comm.allreduce_sum(t)
    torch.cuda.synchronize()
    assert torch.equal(t, want)


@pytest.mark.parametrize("shape", [(10, 10, 22, torch.float32), (20, 20, 64, torch.float32), (12, 9, 5, torch.float64)])
def test_sharded_steps_equal_plain_steps(comm, shape):
    from ark_analysis_amd import som_device
    xd, yd, c, dt = shape
    gen = torch.Generator(device="cpu").manual_seed(5)
    x = torch.rand(6000, c, generator=gen, dtype=torch.float64).to(dt).cuda()
    w0 = torch.rand(xd * yd, c, generator=gen, dtype=torch.float64).cuda()
    out = []
    for use in (None, comm):
        st = som_device.BatchTrainState(x.shape[0], c, xd, yd, 8, x.device)
        st.wbuf[0].copy_(w0)
        som_device.batch_train_steps(x, st, 0, 16, 16, (0.05, 0.01), (3.0, 1.0), comm=use)
        w = torch.empty_like(w0)
        som_device.batch_train_finish(st, 16, 16, (0.05, 0.01), (3.0, 1.0), w)
        torch.cuda.synchronize()
        out.append(w.cpu().numpy())
    # binary64 atomics land in a run-dependent order: the two runs agree to rounding, not bit for bit
    np.testing.assert_allclose(out[0], out[1], rtol=1e-12, atol=0)


def test_exchange_needs_a_bound_communicator():
    from ark_analysis_amd import _capi
    rc = _capi.lib().pxsom_comm_allreduce_sum_f64(None, None, 0, None)
    assert rc != 0 and b"communicator" in _capi.lib().pxsom_last_error()


def test_native_exchange_under_a_one_rank_rccl_group():
    """distributed.native_exchange end to end (bind -> id -> broadcast over the group -> collective create ->
    agreement) in a fresh process with a 1-rank "nccl" process group."""
    code = (
        "import os, torch, torch.distributed as dist\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29577', RANK='0', WORLD_SIZE='1')\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "from ark_analysis_amd import distributed\n"
        "c = distributed.native_exchange(None)\n"
        "assert c is not None and c.nranks == 1, c\n"
        "assert distributed.native_exchange(None) is c\n"
        "t = torch.ones(2300, dtype=torch.float64, device='cuda'); c.allreduce_sum(t); torch.cuda.synchronize()\n"
        "assert float(t.sum()) == 2300.0\n"
        "dist.destroy_process_group()\n"
        "print('EXCHANGE_OK')\n")
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert "EXCHANGE_OK" in res.stdout, res.stdout + res.stderr
