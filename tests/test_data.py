"""CPU: slot / ordering logic of the double-buffered host-batch prefetcher (mtp_amd/data.py); the GPU run is in test_hip_backbone."""
import numpy as np
import pytest
import torch

from mtp_amd.data import HostBatchPrefetcher


def batches(n, shape=(2, 4, 4, 3)):
    for i in range(n):
        yield np.full(shape, i, dtype=np.uint8)


@pytest.mark.parametrize("depth", [2, 3])
@pytest.mark.parametrize("n", [0, 1, 2, 5])
def test_yields_every_batch_once_in_order(depth, n):
    pf = HostBatchPrefetcher(batches(n), device="cpu", depth=depth)
    seen = [int(b[0, 0, 0, 0]) for b in pf]
    assert seen == list(range(n)) and pf.bytes_copied == n * 2 * 4 * 4 * 3


def test_a_batch_stays_valid_until_the_next_call_and_slots_are_recycled():
    pf = HostBatchPrefetcher(batches(6), device="cpu", depth=2)
    a = next(pf)
    assert int(a.max()) == 0 and pf.filled == [True, True]          # batch 1 already staged behind batch 0
    keep = a.clone()
    b = next(pf)                                                       # slot of batch 0 is refilled (with batch 2) only now
    assert torch.equal(keep, torch.zeros_like(keep)) and int(b.max()) == 1
    assert int(a.max()) == 2 and a.data_ptr() == pf.slots[0][1].data_ptr()   # two device buffers in total, reused
    assert len(pf.slots) == 2


def test_rejects_ragged_and_device_batches():
    pf = HostBatchPrefetcher(iter([np.zeros((2, 4, 4, 3), np.uint8), np.zeros((2, 5, 4, 3), np.uint8)]), device="cpu")
    with pytest.raises(ValueError, match="one shape"):
        list(pf)
