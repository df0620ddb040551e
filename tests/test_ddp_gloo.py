"""Data-parallel path on CPU: world_size-2 gloo processes exercise the flat-buffer
gradient reducer (bucketing, range merging, sum + 1/world scale), the rank-0 weight
broadcast over the flat parameter buffer, and rank-local mask streams."""
import os

import pytest
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "vit-ssl_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vitssl_hip.engine import FlatStore, GradReducer
        from vit_core.ssl.simmim import SimMIMViT
        from vit_core.ssl.simmim.masking import draw_mask

        torch.manual_seed(100 + rank)                      # deliberately different init per rank
        model = SimMIMViT(num_blocks=2, input_shape=(3, 32, 32), embed_dim=128, patch_size=8, num_heads=2, mlp_dim=192,
                          dropout=0.0, mask_ratio=0.6)
        store = FlatStore(model, torch.device("cpu"))
        assert store.is_attached()
        # q/k/v adjacency: the fused [3D, D] view exists and aliases the three parameters
        qkv = store.span_view("encoder_blocks.0.self_attention.w_query.weight", "encoder_blocks.0.self_attention.w_value.weight", (384, 128))
        assert torch.equal(qkv[128:256], model.encoder_blocks[0].self_attention.w_key.weight)
        # weights: broadcast rank 0's flat buffer
        dist.broadcast(store.flat, 0)
        ref = [torch.empty_like(store.flat) for _ in range(world)]
        dist.all_gather(ref, store.flat)
        assert all(torch.equal(r, ref[0]) for r in ref)
        assert torch.equal(model.mask_token.data.reshape(-1), store.view("mask_token"))   # Parameters see the broadcast

        # gradients: rank-specific values, handed to the reducer in backward order
        g = store.gflat
        g.copy_(torch.arange(g.numel(), dtype=torch.float32) * (rank + 1))
        red = GradReducer(g, bucket_mb=0.25)               # small buckets -> several collectives
        red.begin()
        order = []
        order.append(store.span("simmim_head.weight", "simmim_head.bias"))
        for i in (1, 0):
            names = [n for n in store.names if n.startswith(f"encoder_blocks.{i}.")]
            order.append(store.span(names[0], names[-1]))
        order.append(store.span("mask_token", "positional_embedding"))
        order.append(store.span("projection.weight", "projection.bias"))
        for lo, hi in order:
            red.ready(lo, hi)
        red.finish()
        covered = torch.zeros(g.numel(), dtype=torch.bool)
        for lo, hi in red.launched:
            assert not covered[lo:hi].any(), "a range was reduced twice"
            covered[lo:hi] = True
        for n in store.names:                                # every parameter's gradient was reduced exactly once
            o, cnt = store.offsets[n]
            assert covered[o:o + cnt].all(), n
        assert len(red.launched) >= 2                        # really bucketed
        want = torch.arange(g.numel(), dtype=torch.float32) * sum(r + 1 for r in range(world))
        for n in store.names:
            o, cnt = store.offsets[n]
            assert torch.equal(g[o:o + cnt], want[o:o + cnt]), n
        assert abs(red.grad_scale - 1.0 / world) < 1e-12

        # rank-local masks: different seeds give different masks, same seed the same
        torch.manual_seed(1000 + rank)
        m = draw_mask(4, 16, 0.6)
        ms = [torch.empty_like(m) for _ in range(world)]
        dist.all_gather(ms, m)
        assert not torch.equal(ms[0], ms[1])
        torch.manual_seed(5)
        m2 = draw_mask(4, 16, 0.6)
        ms2 = [torch.empty_like(m2) for _ in range(world)]
        dist.all_gather(ms2, m2)
        assert torch.equal(ms2[0], ms2[1])
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_single_process_reducer_is_a_noop():
    sys.path.insert(0, os.path.join(ROOT, "vit-ssl_amd"))
    from vitssl_hip.engine import GradReducer
    g = torch.arange(1000, dtype=torch.float32)
    red = GradReducer(g)
    red.begin()
    red.ready(0, 1000)
    red.finish()
    assert torch.equal(g, torch.arange(1000, dtype=torch.float32)) and red.grad_scale == 1.0


def test_reducer_refuses_missing_or_repeated_ranges():
    """finish() checks that one backward handed over every gradient range exactly once."""
    sys.path.insert(0, os.path.join(ROOT, "vit-ssl_amd"))
    from vitssl_hip import VitsslError
    from vitssl_hip.engine import GradReducer
    g = torch.zeros(4096)
    red = GradReducer(g, bucket_mb=0.001)
    red.begin()
    red.ready(2048, 4096)
    red.ready(0, 1024)
    with pytest.raises(VitsslError, match="never handed"):
        red.finish()
    red.begin()
    red.ready(2000, 4096)
    red.ready(1024, 2048)
    red.ready(0, 1024)
    with pytest.raises(VitsslError, match="reduced twice"):
        red.finish()
    red.begin()
    for lo, hi in ((3000, 4096), (1000, 2990), (0, 1000)):      # gaps smaller than the 64-element alignment padding are fine
        red.ready(lo, hi)
    red.finish()
    assert red.stats() == (len(red.launched), 4 * (1096 + 1990 + 1000))
