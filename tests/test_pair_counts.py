"""losses.PairCounts: the host-side bookkeeping of e3dgs_event_loss_cached (which ground-truth pairs have left their
count(D* != 0) on the device).  Pure host logic: CPU tensors stand in for the frames."""
import gc

import torch

from event_3dgs_amd.losses import PairCounts


def test_pairs_are_matched_by_identity_and_in_place_version():
    pc = PairCounts(capacity=3)
    a, b = torch.rand(3, 4, 5), torch.rand(3, 4, 5)
    assert pc.lookup(a, b, 0.17) is None
    cnt = torch.zeros(1, dtype=torch.float64)
    pc.remember(a, b, 0.17, cnt)
    assert pc.lookup(a, b, 0.17) is cnt
    assert pc.lookup(b, a, 0.17) is None                      # the pair is ordered (now, next)
    assert pc.lookup(a, b, 0.2) is None                       # another ground-truth threshold: another D*
    assert pc.lookup(a.clone(), b, 0.17) is None              # equal values, another tensor: not assumed equal
    b.mul_(0.5)                                               # a frame changed in place: the count is stale
    assert pc.lookup(a, b, 0.17) is None
    pc.remember(a, b, 0.17, cnt)
    assert pc.lookup(a, b, 0.17) is cnt


def test_a_recycled_object_id_is_not_a_hit_and_the_cache_is_bounded():
    pc = PairCounts(capacity=2)
    keep = torch.rand(3, 2, 2)
    t = torch.rand(3, 2, 2)
    pc.remember(t, keep, 0.17, torch.zeros(1, dtype=torch.float64))
    key = (id(t), id(keep))
    del t
    gc.collect()
    # whatever object now lives at the old id: the weak reference is dead, the entry cannot match
    assert key in pc._entries and pc._entries[key][0]() is None
    for _ in range(5):
        x = torch.rand(3, 2, 2)
        pc.remember(x, keep, 0.17, torch.zeros(1, dtype=torch.float64))
        assert pc.lookup(x, keep, 0.17) is not None
        assert len(pc._entries) <= 2
