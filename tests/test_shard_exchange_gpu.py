"""amdr_shard_pack_device / amdr_shard_merge_device (csrc/shard.hip): the two launches either side of the all-gather
of the row-sharded corpus (SURVEY.md 8b "amdr_shard_*", 8e).  Checked against oracle/dense.py merge_topk (score desc,
ties -> lower global id, -1 padding) and, bit for bit, against the per-channel torch form + merge_parts_kernel that
rounds 1-3 shipped."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _parts(rng, world, nq, k, dtype, n_rows, ties):
    """Per-rank local top-k lists as a shard search would produce them: sorted descending, local ids, -1 padding at the
    tail of short lists; with `ties`, scores drawn from a handful of values so that equal scores meet across ranks."""
    out = []
    for r in range(world):
        if ties:
            s = rng.integers(0, 6, size=(nq, k)).astype(dtype) * dtype(0.25) - dtype(0.5)
        else:
            s = rng.standard_normal((nq, k)).astype(dtype)
        i = np.stack([rng.choice(n_rows, size=k, replace=False) for _ in range(nq)]).astype(np.int64)
        order = np.lexsort((i, -s.astype(np.float64)), axis=1)  # (score desc, id asc) as every channel returns them
        s, i = np.take_along_axis(s, order, 1), np.take_along_axis(i, order, 1)
        short = rng.integers(0, k + 1, size=nq) if r % 2 else np.full(nq, k)
        for q in range(nq):
            i[q, short[q]:] = -1
            s[q, short[q]:] = -np.finfo(dtype).max
        out.append((s, i))
    return out


@pytest.mark.parametrize("world,ks,ties", [
    (1, (10, 10, 10), False), (2, (10, 10, 10), True), (8, (10, 10, 10), False), (8, (10, 10, 10), True),
    (4, (1, 3, 16), True), (3, (64, 40, 7), False), (8, (80, 80, 80), True), (2, (256, 10, 128), False),
    (20, (10, 6), True), (13, (5,), False), (6, (32, 32, 32, 32), True),
])
def test_pack_and_merge_equal_the_oracle_and_the_torch_form(world, ks, ties):
    from legal_rag_amd import _native
    from legal_rag_amd.retrieval import sharding
    from oracle import dense as OD
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(world * 1000 + sum(ks))
    nq, n_rows = 53, 400
    dts = [np.float32, np.float64, np.float32, np.float64][:len(ks)]
    offsets = [r * n_rows for r in range(world)]
    per_rank = [[_parts(rng, 1, nq, k, dt, n_rows, ties)[0] for k, dt in zip(ks, dts)] for _ in range(world)]
    row = _native.shard_row_words(ks)
    assert row == 2 * sum(ks)
    gathered = torch.empty((world, nq, row), dtype=torch.int64, device=dev)
    for r in range(world):
        chans = [(torch.from_numpy(s).to(dev), torch.from_numpy(i).to(dev)) for s, i in per_rank[r]]
        _native.shard_pack_device([(s.data_ptr(), i.data_ptr(), k, s.dtype == torch.float64) for (s, i), k in zip(chans, ks)],
                                  nq, offsets[r], gathered[r].data_ptr(), device=0,
                                  stream=int(torch.cuda.current_stream().cuda_stream))
        # the wire format is the torch form's: rounds 1-3 packed with where / to / cat
        ref = sharding.pack_channels([(s, sharding.to_global(i, offsets[r])) for s, i in chans])
        assert torch.equal(gathered[r], ref)
    out = [(torch.empty((nq, k), dtype=torch.float64 if dt == np.float64 else torch.float32, device=dev),
            torch.empty((nq, k), dtype=torch.int64, device=dev)) for k, dt in zip(ks, dts)]
    _native.shard_merge_device(gathered.data_ptr(), world, nq,
                               [(s.data_ptr(), i.data_ptr(), k, s.dtype == torch.float64) for (s, i), k in zip(out, ks)],
                               device=0, stream=int(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    unpacked = sharding.unpack_channels(gathered, ks, [o[0].dtype for o in out])
    for c, k in enumerate(ks):
        sp = [per_rank[r][c][0] for r in range(world)]
        ip = [np.where(per_rank[r][c][1] >= 0, per_rank[r][c][1] + offsets[r], -1) for r in range(world)]
        es, ei = OD.merge_topk(sp, ip, k)
        gs, gi = out[c][0].cpu().numpy(), out[c][1].cpu().numpy()
        assert np.array_equal(gi, ei), (c, k)
        valid = ei >= 0
        assert np.array_equal(gs[valid], es[valid]), (c, k)          # score bits
        assert (gs[~valid] == -np.finfo(gs.dtype).max).all()
        # and the old per-channel launch
        os_, oi = sharding.native_merge(unpacked[c][0], unpacked[c][1], k)
        assert torch.equal(oi, out[c][1]) and torch.equal(os_, out[c][0]), (c, k)


def test_exchange_topk_world1_takes_the_native_launches_and_rejects_bad_channels():
    from legal_rag_amd import _native
    from legal_rag_amd.retrieval import sharding
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(3)
    s = torch.sort(torch.randn((7, 10), generator=g), dim=1, descending=True).values.to(dev)
    i = torch.stack([torch.randperm(50, generator=g)[:10] for _ in range(7)]).to(dev)
    b = s.double() * 3
    (gs, gi), (bs, bi) = sharding.exchange_topk([(s, i), (b, i)], 1000)
    torch.cuda.synchronize()
    assert torch.equal(gs, s) and torch.equal(gi, i + 1000) and torch.equal(bs, b) and torch.equal(bi, i + 1000)
    with pytest.raises(_native.NativeError):
        _native.shard_pack_device([(s.data_ptr(), i.data_ptr(), 0, False)], 7, 0, gs.data_ptr())
    with pytest.raises(_native.NativeError):
        _native.shard_merge_device(0, 1, 7, [(gs.data_ptr(), gi.data_ptr(), 10, False)])
