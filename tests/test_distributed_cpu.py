"""CPU, world_size 2 over gloo: the data-parallel path (dl_vqa_amd/distributed.py).

The HIP kernels cannot run here, so each rank's local gradients come from the CPU oracle; what is
under test is the N>1 logic itself: batch sharding, the GLOBAL-batch loss divisor, the bucket
ranges over the flat gradient buffer (every parameter reduced exactly once), SUM all-reduce and
the initial parameter broadcast.  Property: N-rank reduced gradients == 1-rank gradients on the
concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.golden_util import tiny_cfg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_model(seed):
    from dl_vqa_amd import VqaNet

    class CpuFlatVqaNet(VqaNet):          # test-only: keep the flat buffers on the CPU
        def _ensure_flat(self):
            if self._flat_param is None:
                self._flatten(torch.device("cpu"))

    torch.manual_seed(seed)
    cfg = tiny_cfg(dict(bidirectional=True, stride=1, do_option="+"))
    return CpuFlatVqaNet(cfg, 40), cfg


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dl_vqa_amd.distributed import GROUPS, DataParallel, shard_batch
        from oracle import vqa_oracle as O
        # ranks start from DIFFERENT weights: the wrapper must broadcast rank 0's
        model, cfg = _make_model(seed=100 + rank)
        dp = DataParallel(model)
        assert model._seed_rank == rank and model._grad_sync is dp
        btot = 3 * world
        batch = O.synthetic_batch(btot, 32, 5, 40, 12, seed=9)
        v, q, a_idx, a_val, _, _, q_len = shard_batch(batch, rank, world)
        assert v.shape[0] == 3
        sd = {k: p.data.clone() for k, p in model.named_parameters()}
        _, _, grads = O.loss_and_grads(sd, cfg, v, q, q_len, a_idx, a_val, loss_scale_batch=btot)
        views = model._grad_views()
        for k, g in grads.items():
            views[k].copy_(g)
        for group in GROUPS:                      # the order backward produces them
            dp.bucket_ready(model, group)
        dp.finish(model)
        torch.save({"flat_grad": model._flat_grad.clone(), "flat_param": model._flat_param.clone()},
                   os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_two_rank_gradients_equal_single_rank_on_concatenated_batch(tmp_path, world):
    """world_size 2 and 4 (the driver's scaling runs go to 8): every rank ends with rank 0's parameters and the gradient of
    the concatenated batch."""
    from oracle import vqa_oracle as O
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    for k in range(1, world):
        rk = torch.load(tmp_path / f"rank{k}.pt")
        assert torch.equal(r0["flat_param"], rk["flat_param"])      # broadcast made the replicas equal
        assert torch.equal(r0["flat_grad"], rk["flat_grad"])        # all-reduce: same result everywhere
    model, cfg = _make_model(seed=100)                               # rank 0's weights
    model._ensure_flat()
    assert torch.equal(model._flat_param, r0["flat_param"])
    v, q, a_idx, a_val, _, _, q_len = O.synthetic_batch(3 * world, 32, 5, 40, 12, seed=9)
    sd = {k: p.data.clone() for k, p in model.named_parameters()}
    _, _, grads = O.loss_and_grads(sd, cfg, v, q, q_len, a_idx, a_val)
    views = model._grad_views()
    covered = torch.zeros_like(model._flat_grad, dtype=torch.bool)
    for k, g in grads.items():
        o, n = model._offsets[k]
        got = r0["flat_grad"][o:o + n].view(g.shape)
        scale = max(float(g.abs().max()), 1e-12)
        assert float((got - g).abs().max()) / scale < 2e-4 or k == "attention.x_conv.bias", k
        covered[o:o + n] = True
    # every parameter lies in exactly one bucket
    from dl_vqa_amd.distributed import GROUPS
    seen = torch.zeros_like(covered, dtype=torch.int32)
    for gname in GROUPS:
        lo, hi = model.group_range(gname)
        seen[lo:hi] += 1
    assert bool((seen[covered] == 1).all())


def test_shard_batch_rejects_uneven_split():
    from dl_vqa_amd.distributed import shard_batch
    with pytest.raises(AssertionError):
        shard_batch((torch.zeros(5, 3), torch.zeros(5)), 0, 2)
