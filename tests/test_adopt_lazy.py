"""CPU: the lazy tensors behind adopt.render's deferred renders (event_3dgs_amd/adopt.py: _Lazy, _PendingRenders) -- the
mechanics that need no GPU: metadata without rendering, every torch operation sees the real tensor (with its autograd
history), a first use under no_grad still builds the graph, indexing with a lazy mask, materialize()."""
import pytest
import torch


class _Batch:
    """Stands in for _PendingRenders: counts flushes, builds its value with gradients enabled as the real one does."""

    def __init__(self, leaf):
        self.leaf, self.flushes, self.value = leaf, 0, None

    def flush(self):
        if self.value is None:
            self.flushes += 1
            with torch.enable_grad():
                self.value = self.leaf * 2.0


def _lazy(batch, shape=(2, 3), dtype=torch.float32):
    from event_3dgs_amd import adopt
    return adopt._Lazy(batch, lambda: batch.value, shape, dtype, torch.device("cpu"))


def test_metadata_is_answered_without_rendering():
    b = _Batch(torch.arange(6.0).reshape(2, 3).requires_grad_(True))
    x = _lazy(b)
    assert tuple(x.shape) == (2, 3) and x.dtype == torch.float32 and x.device.type == "cpu" and not x.is_cuda
    assert x.dim() == 2 and x.numel() == 6 and x.size(1) == 3 and len(x) == 2 and x.is_floating_point()
    assert isinstance(x, torch.Tensor) and b.flushes == 0


def test_operations_run_on_the_real_tensor_and_carry_its_history():
    from event_3dgs_amd import adopt
    leaf = torch.arange(6.0).reshape(2, 3).requires_grad_(True)
    b = _Batch(leaf)
    x = _lazy(b)
    loss = torch.abs(x - 1.0).mean() + (x * x).sum() + x[0, 1]         # function, operators, indexing
    assert b.flushes == 1 and loss.grad_fn is not None
    loss.backward()
    real = leaf.detach() * 2.0
    expect = 2.0 * (torch.sign(real - 1.0) / 6.0 + 2.0 * real)
    expect[0, 1] += 2.0
    assert torch.allclose(leaf.grad, expect)
    assert adopt.materialize(x) is b.value and adopt.materialize(leaf) is leaf
    assert x.detach().numpy().shape == (2, 3) and float(x.sum()) == float(real.sum())


def test_first_use_under_no_grad_still_builds_the_graph():
    leaf = torch.ones(2, 3, requires_grad=True)
    b = _Batch(leaf)
    x = _lazy(b)
    with torch.no_grad():
        assert float(x.mean()) == 2.0                                   # a logging line
    (x.sum()).backward()                                                # the value built there has its history
    assert torch.equal(leaf.grad, torch.full((2, 3), 2.0))


def test_a_lazy_mask_as_an_index_and_in_place_updates_through_it():
    # train.py:318-320: max_radii2D[visibility_filter] = torch.max(max_radii2D[visibility_filter], radii[visibility_filter])
    class B:
        def flush(self):
            pass
    from event_3dgs_amd import adopt
    radii = adopt._Lazy(B(), lambda: torch.tensor([3, 0, 7, 0], dtype=torch.int32), (4,), torch.int32, torch.device("cpu"))
    vis = adopt._Lazy(B(), lambda: torch.tensor([True, False, True, False]), (4,), torch.bool, torch.device("cpu"))
    acc = torch.tensor([5.0, 5.0, 5.0, 5.0])
    acc[vis] = torch.max(acc[vis], radii[vis].float())
    assert acc.tolist() == [5.0, 5.0, 7.0, 5.0]
    cnt = torch.zeros(4)
    cnt[vis] += 1
    assert cnt.tolist() == [1.0, 0.0, 1.0, 0.0]


def test_pending_renders_refuse_parameters_that_moved():
    """_PendingRenders.flush() compares the parameter versions it was issued on (no rasteriser involved here)."""
    from event_3dgs_amd import adopt

    class PC:
        pass
    pc = PC()
    for name in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        setattr(pc, name, torch.zeros(4, 3))

    class RS:
        bg = torch.zeros(3); scale_modifier = 1.0; sh_degree = 3; image_height = 8; image_width = 8
    batch = adopt._PendingRenders(pc, adopt._PendingRenders.key_of(pc, RS))
    pc._xyz.add_(1.0)                                                   # an optimizer step between render() and first use
    with pytest.raises(RuntimeError, match="modified in place"):
        batch.flush()
