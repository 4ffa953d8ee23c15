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


def test_threaded_walk_equals_serial(lp):
    """pack(..., nthreads): the outermost dimension dealt to native threads that run without the GIL; same result, and
    everything the threads do not expect (numpy scalars, long ints, ragged lists) falls back to the serial walk, which
    handles it or raises as before"""
    rng = np.random.default_rng(4)
    shape = (64, 11, 10, 4)                       # 28,160 elements: below the size where threads start ...
    big = (256, 11, 10, 4)                        # ... and above it
    for sh in (shape, big):
        a = rng.integers(-3, 1_600_000, sh)
        nested = a.tolist()
        nested[1][2] = np.zeros(sh[2:]).tolist()                 # float dummies
        nested[5][0][3][1] = 7.9                                  # truncated like ndarray.astype(int32)
        want = np.asarray(nested).astype(np.int32)
        for nt in (1, 2, 3, 8, 64):
            out = np.full(want.size + 2, -7, dtype=np.int32)
            lp.pack(nested, out, sh, nt)
            assert np.array_equal(out[:want.size].reshape(sh), want) and (out[want.size:] == -7).all(), (sh, nt)
    a = rng.integers(0, 100, big)
    nested = a.tolist()
    nested[200][3][4][2] = np.int64(55)                          # not an exact int: the serial walk converts it
    out = np.zeros(a.size, dtype=np.int32)
    lp.pack(nested, out, big, 4)
    a[200, 3, 4, 2] = 55
    assert np.array_equal(out.reshape(big), a)
    nested[17][0][0][0] = 2 ** 40
    with pytest.raises(OverflowError):
        lp.pack(nested, out, big, 4)
    nested[17][0][0][0] = 1
    nested[255][10] = nested[255][10][:-1]                       # ragged in the last thread's share
    with pytest.raises(ValueError):
        lp.pack(nested, out, big, 4)
    assert lp.NOGIL_THREADS in (0, 1)


def test_pack_many_equals_pack_per_tensor(lp):
    """pack_many: the whole feed tuple in one call (one threaded region: the threads are created once); results and errors
    are those of pack() on each tensor, and the exception says which tensor it was"""
    rng = np.random.default_rng(6)
    B, T, K = 256, 11, 10
    arrs = [rng.integers(-2, 1_500_000, (B, T, K, 4)), rng.integers(0, 99, (B, T, K, 3)), rng.integers(0, 9, (B, 3)),
            rng.integers(0, 2, (B,))]
    lists = [a.tolist() for a in arrs]
    lists[0][3][2] = np.zeros((K, 4)).tolist()
    want = [np.asarray(l).astype(np.int32) for l in lists]
    for nt in (1, 3, 8, 32):
        outs = [np.full(a.size, -7, np.int32) for a in arrs]
        lp.pack_many([(l, o, tuple(a.shape)) for l, o, a in zip(lists, outs, arrs)], nt)
        assert all(np.array_equal(o.reshape(a.shape), w) for o, a, w in zip(outs, arrs, want)), nt
    outs = [np.zeros(a.size, np.int32) for a in arrs]
    ragged = list(lists)
    ragged[1] = ragged[1][:-1]
    with pytest.raises(ValueError) as ei:
        lp.pack_many([(l, o, tuple(a.shape)) for l, o, a in zip(ragged, outs, arrs)], 4)
    assert ei.value.tensor_index == 1
    lists[0][200][5][5][1] = 2 ** 40
    with pytest.raises(OverflowError) as ei:
        lp.pack_many([(l, o, tuple(a.shape)) for l, o, a in zip(lists, outs, arrs)], 4)
    assert ei.value.tensor_index == 0
    with pytest.raises(ValueError):
        lp.pack_many([(lists[3], np.zeros(3, np.int32), (B,))], 2)          # buffer too small
