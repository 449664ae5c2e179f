"""cannon.gather_blocks: blocks of a flat buffer back to back in a new order (the packing / sorting step of make_images),
including the chunked path."""
import numpy as np
import torch

from dbcsr_amd import cannon


def test_gather_blocks_matches_numpy(monkeypatch):
    rng = np.random.default_rng(5)
    lens = rng.integers(0, 40, size=300).astype(np.int64)
    starts = np.concatenate([[0], np.cumsum(lens)])[:-1].astype(np.int64)
    src = torch.from_numpy(rng.standard_normal(int(lens.sum())))
    perm = rng.permutation(len(lens))
    want = np.concatenate([src.numpy()[starts[b]:starts[b] + lens[b]] for b in perm])
    for chunk in (1 << 26, 97, 1):
        monkeypatch.setattr(cannon, "_GATHER_CHUNK", chunk)
        got = cannon.gather_blocks(src, starts[perm], lens[perm])
        assert np.array_equal(got.numpy(), want)
    assert cannon.gather_blocks(src, np.zeros(0, np.int64), np.zeros(0, np.int64)).numel() == 0
