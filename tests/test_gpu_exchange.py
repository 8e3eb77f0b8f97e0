"""-m gpu: the exchange step of the batch-sharded vamp() on the DEVICE collective (RCCL), as far as a one-GPU box can run it: a
one-rank "nccl" process group.  Interface._allgather_batch runs whenever a process group was given, so the same code — the padded
local block, dist.all_gather_into_tensor on device tensors through RCCL, the trim — executes here as on the eight GPUs of a node
(world_size 2 with ragged shards is covered on CPU with gloo: tests/test_interface_host.py)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(port, q, exchange="torch"):
    import torch.distributed as dist
    from oracle import vampnet_oracle as O, weights as W
    from tests.gpu_common import SynthCodec, model_kwargs
    from vampnet_amd.interface import Interface
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))      # "nccl" is RCCL on ROCm
    try:
        cb = W.synth_codebooks()
        csd, fsd = W.synth_state_dict(W.TINY_COARSE_DIMS, 0), W.synth_state_dict(W.TINY_C2F_DIMS, 1)
        itf = Interface.from_state_dicts(SynthCodec(cb), csd, model_kwargs(W.TINY_COARSE_DIMS), fsd, model_kwargs(W.TINY_C2F_DIMS),
                                         device="cuda:0", max_batch=3, process_group=dist.group.WORLD, exchange=exchange)
        itf.exchange_log = []
        z = W.synth_codes(3, 14, 200, seed=6)
        torch.manual_seed(3)
        mask = itf.build_mask(z)
        got = itf.vamp(z, mask, batch_size=3, seed=11, _sampling_steps=3).cpu()
        ref = O.vamp(O.OracleModels(csd, W.TINY_COARSE_DIMS, fsd, W.TINY_C2F_DIMS, cb), z, mask, batch_size=3, seed=11, _sampling_steps=3)
        # the raw collective on a block the trim has to cut (5 rows in a 1-rank group: per = 5, no padding possible with one rank;
        # the padded form runs in the 2-rank gloo test)
        t = torch.arange(5 * 14 * 7, device="cuda").reshape(5, 14, 7)
        same = torch.equal(itf._allgather_batch(t), t)
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in itf.exchange_log]
        q.put(dict(ok=bool(torch.equal(got, ref)), n_exchange=len(ms), ms=ms, backend=dist.get_backend(), raw=same))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["torch", "c_abi"])
def test_vamp_exchange_on_one_rank_rccl_group(exchange):
    """exchange="torch": torch.distributed's all_gather_into_tensor on the nccl (= RCCL) group; "c_abi": the library's own RCCL
    communicator behind vn_comm_create / vn_allgather_tokens (include/vampnet_hip.h), the unique id carried by the group"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q, exchange))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    print(f"one-rank RCCL exchange [{exchange}]:", res)
    assert res["backend"] == "nccl"
    assert res["ok"], "vamp() through the RCCL exchange differs from the oracle"
    assert res["n_exchange"] == 2 and res["raw"]                 # one all-gather per vamp() + the raw call
    assert all(m >= 0.0 for m in res["ms"])
