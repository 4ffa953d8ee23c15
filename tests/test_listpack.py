"""The nested-list ingestion helper (score_amd/cext/listpack.c): the feed_dict conversion of score.py:102-115 for what
GraphLoader yields (graph_loader.py:383: nested lists of ints with float 0.0 dummies, :90-91), against NumPy's."""
import numpy as np
import pytest

from score_amd import _lib


@pytest.fixture(scope="module")
def lp():
    m = _lib.listpack()
    assert m is not None, "the _listpack extension must build (gcc + Python.h are part of the image)"
    return m


def test_pack_equals_numpy_conversion(lp):
    rng = np.random.default_rng(0)
    for shape in ((7,), (5, 3), (4, 3, 2, 4), (3, 11, 10, 3)):
        a = rng.integers(0, 1_600_000, shape)
        nested = a.tolist()
        if len(shape) == 4:                       # dummy slices arrive as float zeros (graph_loader.py:90-91)
            nested[0][0] = np.zeros(shape[2:]).tolist()
            nested[-1][-1] = np.zeros(shape[2:]).tolist()
        want = np.asarray(nested).astype(np.int32)
        out = np.full(want.size + 3, -7, dtype=np.int32)
        lp.pack(nested, out, tuple(shape))
        assert np.array_equal(out[:want.size].reshape(shape), want) and (out[want.size:] == -7).all()
    out = np.zeros(6, dtype=np.int32)
    lp.pack(((1, 2.9, -3.9), [np.int64(4), True, np.float32(6.0)]), out, (2, 3))      # tuples, truncation, numpy scalars
    assert out.tolist() == [1, 2, -3, 4, 1, 6]


def test_pack_rejects_what_does_not_fit(lp):
    out = np.zeros(8, dtype=np.int32)
    with pytest.raises(ValueError):
        lp.pack([[1, 2], [3]], out, (2, 2))                   # ragged
    with pytest.raises(ValueError):
        lp.pack([1, 2, 3], out, (2,))                         # wrong length
    with pytest.raises(ValueError):
        lp.pack(5, out, (1,))                                 # not a sequence
    with pytest.raises(OverflowError):
        lp.pack([1, 2 ** 40], out, (2,))
    with pytest.raises(OverflowError):
        lp.pack([1e12], out, (1,))
    with pytest.raises(ValueError):
        lp.pack([1, 2, 3], np.zeros(2, dtype=np.int32), (3,))   # buffer too small
    with pytest.raises((TypeError, ValueError)):
        lp.pack([1, "x"], out, (2,))


def test_pack_is_what_the_model_feed_path_uses():
    # (the GPU test test_nested_list_feed_and_errors runs the whole path; here: the staging buffer on the host)
    from score_amd.model import batch_shapes, carve_batch, flat_batch_size
    import torch

    class Cfg(object):
        max_time_len, obj_per_time_slice, user_fnum, item_fnum = 3, 2, 3, 4
    shapes = batch_shapes(Cfg, 5)
    assert shapes[0] == (5, 3, 2, 4) and shapes[4] == (5, 3) and shapes[7] == (5,)
    flat = torch.zeros(flat_batch_size(shapes), dtype=torch.int32)
    views = carve_batch(flat, shapes)
    assert [tuple(v.shape) for v in views] == list(shapes)
    assert all(v.data_ptr() % 16 == 0 for v in views)          # every tensor 16-B aligned inside the flat buffer
