/* _listpack: flatten the nested Python lists of a feed tuple into int32, in C.
 *
 * The reference's loader hands model.train / model.eval nested lists (graph_loader.py:383) of Python ints, with float 0.0
 * in dummy slices (:90-91), and `sess.run(feed_dict=...)` converts them (score.py:102-115).  np.asarray on such lists
 * costs ~70 ns per element: 22 ms for the 309,800 ids of a B = 200 Tmall batch -- forty times the whole training step
 * on the GPU.  This walks the lists with the CPython list API instead (~5 ns per element) and writes straight into
 * the flat int32 staging buffer of score_amd.model.DeviceBatch.
 *
 *   pack(obj, out, shape, nthreads=1) -> None
 *     obj    nested lists / tuples whose nesting matches `shape` (a tuple of ints); leaves: int (or anything with
 *            __index__ / __int__), float (truncated toward zero, as ndarray.astype(int32) does)
 *     out    writable C-contiguous buffer of int32 with at least prod(shape) elements
 *     nthreads > 1: the walk is memory-latency bound (every boxed int is a cache miss: 3.8 ns per element, 10.8 ms for
 *            the 2.87 M ids of a cfg-3 batch), so the outermost dimension is dealt to `nthreads` native threads that
 *            run WITHOUT the GIL.  They only read immutable fields (types, sizes, the digit of an exact int, the value
 *            of an exact float) and never touch a reference count or the error state; anything they do not expect --
 *            another leaf type, a long int, a wrong length -- makes the call fall back to the serial walk below,
 *            which handles it or raises.  The caller must not mutate `obj` from another thread during the call (the
 *            reference hands over a fresh object unpickled from a queue, graph_loader.py:397-398).
 * Raises ValueError on a shape mismatch, OverflowError on values outside int32.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <pthread.h>
#include <stdint.h>

static int leaf(PyObject* o, int32_t* dst) {
  long v;
  if (PyLong_CheckExact(o)) {
    int ovf = 0;
    v = PyLong_AsLongAndOverflow(o, &ovf);
    if (ovf) { PyErr_SetString(PyExc_OverflowError, "feed value outside int32"); return -1; }
  } else if (PyFloat_CheckExact(o)) {
    double d = PyFloat_AS_DOUBLE(o);
    if (!(d > -2147483649.0 && d < 2147483648.0)) { PyErr_SetString(PyExc_OverflowError, "feed value outside int32"); return -1; }
    v = (long)d;
  } else {
    PyObject* n = PyNumber_Long(o);
    if (!n) return -1;
    int ovf = 0;
    v = PyLong_AsLongAndOverflow(n, &ovf);
    Py_DECREF(n);
    if (ovf) { PyErr_SetString(PyExc_OverflowError, "feed value outside int32"); return -1; }
    if (v == -1 && PyErr_Occurred()) return -1;
  }
  if (v < INT32_MIN || v > INT32_MAX) { PyErr_SetString(PyExc_OverflowError, "feed value outside int32"); return -1; }
  *dst = (int32_t)v;
  return 0;
}

static int walk(PyObject* o, const Py_ssize_t* shape, int nd, int32_t** dst) {
  if (nd == 0) {
    if (leaf(o, *dst) < 0) return -1;
    ++*dst;
    return 0;
  }
  Py_ssize_t n;
  PyObject** items;
  if (PyList_CheckExact(o)) { n = PyList_GET_SIZE(o); items = ((PyListObject*)o)->ob_item; }
  else if (PyTuple_CheckExact(o)) { n = PyTuple_GET_SIZE(o); items = ((PyTupleObject*)o)->ob_item; }
  else { PyErr_SetString(PyExc_ValueError, "nested feed: expected a list or tuple"); return -1; }
  if (n != shape[0]) {
    PyErr_Format(PyExc_ValueError, "nested feed: a list of length %zd where the shape says %zd", n, shape[0]);
    return -1;
  }
  if (nd == 1) {                                   /* innermost lists: the hot loop */
    int32_t* d = *dst;
    for (Py_ssize_t i = 0; i < n; ++i) {
      PyObject* it = items[i];
      if (i + 4 < n) __builtin_prefetch(items[i + 4]);      /* boxed ints are scattered over the heap */
      if (PyLong_CheckExact(it)) {
#if PY_VERSION_HEX < 0x030C0000
        /* CPython < 3.12: non-negative ints below 2^30 are one 30-bit digit -- every row id of the path */
        const Py_ssize_t sz = Py_SIZE(it);
        if (sz == 1) { d[i] = (int32_t)((PyLongObject*)it)->ob_digit[0]; continue; }
        if (sz == 0) { d[i] = 0; continue; }
#endif
        int ovf = 0;
        long v = PyLong_AsLongAndOverflow(it, &ovf);
        if (ovf || v < INT32_MIN || v > INT32_MAX) { PyErr_SetString(PyExc_OverflowError, "feed value outside int32"); return -1; }
        d[i] = (int32_t)v;
      } else if (PyFloat_CheckExact(it) && PyFloat_AS_DOUBLE(it) == 0.0) {
        d[i] = 0;                                   /* the dummy node's float zeros (graph_loader.py:90-91) */
      } else {
        if (leaf(it, d + i) < 0) return -1;
        /* leaf() may have run arbitrary __int__ / __index__ code: the list may have been resized under us */
        if (PyList_CheckExact(o)) {
          if (PyList_GET_SIZE(o) != n) { PyErr_SetString(PyExc_ValueError, "nested feed: a list changed size during the walk"); return -1; }
          items = ((PyListObject*)o)->ob_item;
        }
      }
    }
    *dst += n;
    return 0;
  }
  for (Py_ssize_t i = 0; i < n; ++i) {
    if (i + 1 < n) {
      __builtin_prefetch(items[i + 1]);
      if (nd == 2 && PyList_CheckExact(items[i + 1])) __builtin_prefetch(((PyListObject*)items[i + 1])->ob_item);
    }
    if (walk(items[i], shape + 1, nd - 1, dst) < 0) return -1;
  }
  return 0;
}

/* ---- the same walk without the GIL (see the header): 0 = done, 1 = something the serial walk has to look at ---- */
#if PY_VERSION_HEX < 0x030C0000
#define LISTPACK_NOGIL 1
static int walk_nogil(PyObject* o, const Py_ssize_t* shape, int nd, int32_t* dst, Py_ssize_t stride) {
  Py_ssize_t n;
  PyObject** items;
  if (Py_TYPE(o) == &PyList_Type) { n = Py_SIZE(o); items = ((PyListObject*)o)->ob_item; }
  else if (Py_TYPE(o) == &PyTuple_Type) { n = Py_SIZE(o); items = ((PyTupleObject*)o)->ob_item; }
  else return 1;
  if (n != shape[0]) return 1;
  if (nd == 1) {
    for (Py_ssize_t i = 0; i < n; ++i) {
      PyObject* it = items[i];
      if (i + 4 < n) __builtin_prefetch(items[i + 4]);
      if (Py_TYPE(it) == &PyLong_Type) {
        const Py_ssize_t sz = Py_SIZE(it);
        if (sz == 1) dst[i] = (int32_t)((PyLongObject*)it)->ob_digit[0];
        else if (sz == 0) dst[i] = 0;
        else if (sz == -1) dst[i] = -(int32_t)((PyLongObject*)it)->ob_digit[0];
        else return 1;
      } else if (Py_TYPE(it) == &PyFloat_Type) {
        const double d = PyFloat_AS_DOUBLE(it);
        if (!(d > -2147483649.0 && d < 2147483648.0)) return 1;
        dst[i] = (int32_t)d;
      } else {
        return 1;
      }
    }
    return 0;
  }
  const Py_ssize_t sub = stride / shape[0];      /* elements below one item of this level */
  for (Py_ssize_t i = 0; i < n; ++i) {
    if (i + 1 < n) {
      __builtin_prefetch(items[i + 1]);
      if (nd == 2 && Py_TYPE(items[i + 1]) == &PyList_Type) __builtin_prefetch(((PyListObject*)items[i + 1])->ob_item);
    }
    if (walk_nogil(items[i], shape + 1, nd - 1, dst + i * sub, sub)) return 1;
  }
  return 0;
}
typedef struct { PyObject** items; const Py_ssize_t* shape; int nd; int32_t* dst; Py_ssize_t lo, hi, sub; int status; } lp_job;
static void* lp_thread(void* arg) {
  lp_job* j = (lp_job*)arg;
  j->status = 0;
  for (Py_ssize_t i = j->lo; i < j->hi && !j->status; ++i)
    j->status = walk_nogil(j->items[i], j->shape + 1, j->nd - 1, j->dst + i * j->sub, j->sub);
  return NULL;
}
/* 0 = packed, 1 = fall back to the serial walk */
static int pack_threads(PyObject* o, const Py_ssize_t* shape, int nd, int32_t* dst, Py_ssize_t total, int nthreads) {
  if (nd < 2 || total < 65536) return 1;
  Py_ssize_t n;
  PyObject** items;
  if (PyList_CheckExact(o)) { n = PyList_GET_SIZE(o); items = ((PyListObject*)o)->ob_item; }
  else if (PyTuple_CheckExact(o)) { n = PyTuple_GET_SIZE(o); items = ((PyTupleObject*)o)->ob_item; }
  else return 1;
  if (n != shape[0] || n == 0) return 1;
  if (nthreads > 16) nthreads = 16;
  if (nthreads > n) nthreads = (int)n;
  lp_job jobs[16];
  pthread_t th[16];
  int started[16];
  const Py_ssize_t sub = total / n;
  int bad = 0;
  Py_BEGIN_ALLOW_THREADS
  for (int t = 0; t < nthreads; ++t) {
    lp_job* j = &jobs[t];
    j->items = items; j->shape = shape; j->nd = nd; j->dst = dst; j->sub = sub;
    j->lo = n * t / nthreads; j->hi = n * (t + 1) / nthreads; j->status = 0;
    started[t] = 0;
    if (t > 0) started[t] = pthread_create(&th[t], NULL, lp_thread, j) == 0;
  }
  lp_thread(&jobs[0]);
  for (int t = 1; t < nthreads; ++t) {
    if (started[t]) pthread_join(th[t], NULL);
    else lp_thread(&jobs[t]);                   /* (thread creation failed: do its share here) */
  }
  for (int t = 0; t < nthreads; ++t) bad |= jobs[t].status;
  Py_END_ALLOW_THREADS
  return bad;
}
/* several tensors in ONE threaded region: the threads are created once for the whole feed tuple (thread creation, ~30 us
 * each, was most of what eight separate pack() calls with 8 threads cost) and every thread takes its share of the
 * outermost dimension of every tensor.  status[i] = 0 packed / 1 needs the serial walk. */
#define LP_MAXT 16
typedef struct { PyObject** items; const Py_ssize_t* shape; int nd; int32_t* dst; Py_ssize_t n, sub; int status[LP_MAXT]; } lp_tensor;
typedef struct { lp_tensor* t; int nt, tid, nthreads; } lp_multi;
static void* lp_multi_thread(void* arg) {
  lp_multi* m = (lp_multi*)arg;
  for (int i = 0; i < m->nt; ++i) {
    lp_tensor* t = &m->t[i];
    const Py_ssize_t lo = t->n * m->tid / m->nthreads, hi = t->n * (m->tid + 1) / m->nthreads;
    int st = 0;
    for (Py_ssize_t r = lo; r < hi && !st; ++r) st = walk_nogil(t->items[r], t->shape + 1, t->nd - 1, t->dst + r * t->sub, t->sub);
    t->status[m->tid] = st;
  }
  return NULL;
}
static void pack_many_threads(lp_tensor* t, int nt, int nthreads) {
  if (nthreads > LP_MAXT) nthreads = LP_MAXT;
  lp_multi jobs[LP_MAXT];
  pthread_t th[LP_MAXT];
  int started[LP_MAXT];
  Py_BEGIN_ALLOW_THREADS
  for (int k = 0; k < nthreads; ++k) {
    jobs[k].t = t; jobs[k].nt = nt; jobs[k].tid = k; jobs[k].nthreads = nthreads;
    started[k] = 0;
    if (k > 0) started[k] = pthread_create(&th[k], NULL, lp_multi_thread, &jobs[k]) == 0;
  }
  lp_multi_thread(&jobs[0]);
  for (int k = 1; k < nthreads; ++k) {
    if (started[k]) pthread_join(th[k], NULL);
    else lp_multi_thread(&jobs[k]);
  }
  Py_END_ALLOW_THREADS
}
#else
#define LISTPACK_NOGIL 0
static int pack_threads(PyObject* o, const Py_ssize_t* shape, int nd, int32_t* dst, Py_ssize_t total, int nthreads) { return 1; }
#endif

static PyObject* pack(PyObject* self, PyObject* args) {
  PyObject *obj, *shape_o;
  Py_buffer out;
  int nthreads = 1;
  if (!PyArg_ParseTuple(args, "Ow*O!|i", &obj, &out, &PyTuple_Type, &shape_o, &nthreads)) return NULL;
  PyObject* res = NULL;
  Py_ssize_t shape[8];
  const Py_ssize_t nd = PyTuple_GET_SIZE(shape_o);
  Py_ssize_t total = 1;
  if (nd < 1 || nd > 8) { PyErr_SetString(PyExc_ValueError, "shape must have 1..8 dimensions"); goto done; }
  for (Py_ssize_t i = 0; i < nd; ++i) {
    shape[i] = PyLong_AsSsize_t(PyTuple_GET_ITEM(shape_o, i));
    if (shape[i] < 0) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "negative dimension"); goto done; }
    total *= shape[i];
  }
  if (!PyBuffer_IsContiguous(&out, 'C') || out.len < (Py_ssize_t)(total * sizeof(int32_t)) || ((uintptr_t)out.buf & 3)) {
    PyErr_SetString(PyExc_ValueError, "out must be a C-contiguous, 4-byte aligned buffer of at least prod(shape) int32");
    goto done;
  }
  {
    int32_t* dst = (int32_t*)out.buf;
    if (nthreads <= 1 || pack_threads(obj, shape, (int)nd, dst, total, nthreads) != 0)
      if (walk(obj, shape, (int)nd, &dst) < 0) goto done;
  }
  res = Py_None;
  Py_INCREF(res);
done:
  PyBuffer_Release(&out);
  return res;
}

/* pack_many(items, nthreads) -- items: a sequence of (obj, out, shape) triples as pack() takes them.  Same results and
 * errors as calling pack() on each in turn; the large tensors share one threaded region. */
static PyObject* pack_many(PyObject* self, PyObject* args) {
  PyObject* seq;
  int nthreads = 1;
  if (!PyArg_ParseTuple(args, "O|i", &seq, &nthreads)) return NULL;
  PyObject* fast = PySequence_Fast(seq, "pack_many: items must be a sequence of (obj, out, shape)");
  if (!fast) return NULL;
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
  if (n > 16) { Py_DECREF(fast); PyErr_SetString(PyExc_ValueError, "pack_many: at most 16 tensors"); return NULL; }
  Py_buffer bufs[16];
  Py_ssize_t shapes[16][8];
  int nds[16];
  Py_ssize_t totals[16];
  PyObject* objs[16];
  Py_ssize_t got = 0;
  PyObject* res = NULL;
  for (; got < n; ++got) {
    PyObject* it = PySequence_Fast_GET_ITEM(fast, got);
    PyObject *shape_o, *out_o;
    if (!PyTuple_Check(it) || PyTuple_GET_SIZE(it) != 3) { PyErr_SetString(PyExc_ValueError, "pack_many: item must be (obj, out, shape)"); goto done; }
    objs[got] = PyTuple_GET_ITEM(it, 0); out_o = PyTuple_GET_ITEM(it, 1); shape_o = PyTuple_GET_ITEM(it, 2);
    if (!PyTuple_Check(shape_o)) { PyErr_SetString(PyExc_ValueError, "shape must be a tuple"); goto done; }
    if (PyObject_GetBuffer(out_o, &bufs[got], PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) < 0) goto done;
    const Py_ssize_t nd = PyTuple_GET_SIZE(shape_o);
    if (nd < 1 || nd > 8) { PyBuffer_Release(&bufs[got]); PyErr_SetString(PyExc_ValueError, "shape must have 1..8 dimensions"); goto done; }
    Py_ssize_t total = 1;
    for (Py_ssize_t i = 0; i < nd; ++i) {
      shapes[got][i] = PyLong_AsSsize_t(PyTuple_GET_ITEM(shape_o, i));
      if (shapes[got][i] < 0) { PyBuffer_Release(&bufs[got]); if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "negative dimension"); goto done; }
      total *= shapes[got][i];
    }
    if (bufs[got].len < (Py_ssize_t)(total * sizeof(int32_t)) || ((uintptr_t)bufs[got].buf & 3)) {
      PyBuffer_Release(&bufs[got]);
      PyErr_SetString(PyExc_ValueError, "out must be a C-contiguous, 4-byte aligned buffer of at least prod(shape) int32");
      goto done;
    }
    nds[got] = (int)nd; totals[got] = total;
  }
  {
    int threaded[16];
    for (Py_ssize_t i = 0; i < n; ++i) threaded[i] = 0;
#if LISTPACK_NOGIL
    if (nthreads > 1) {
      lp_tensor ts[16];
      int map[16], nt = 0;
      for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject* o = objs[i];
        Py_ssize_t len;
        PyObject** items;
        if (nds[i] < 2 || totals[i] < 65536) continue;
        if (PyList_CheckExact(o)) { len = PyList_GET_SIZE(o); items = ((PyListObject*)o)->ob_item; }
        else if (PyTuple_CheckExact(o)) { len = PyTuple_GET_SIZE(o); items = ((PyTupleObject*)o)->ob_item; }
        else continue;
        if (len != shapes[i][0] || len == 0) continue;
        ts[nt].items = items; ts[nt].shape = shapes[i]; ts[nt].nd = nds[i]; ts[nt].dst = (int32_t*)bufs[i].buf;
        ts[nt].n = len; ts[nt].sub = totals[i] / len;
        map[nt++] = (int)i;
      }
      if (nt > 0) {
        const int use = nthreads > LP_MAXT ? LP_MAXT : nthreads;
        pack_many_threads(ts, nt, use);
        for (int k = 0; k < nt; ++k) {
          int bad = 0;
          for (int q = 0; q < use; ++q) bad |= ts[k].status[q];
          threaded[map[k]] = !bad;
        }
      }
    }
#endif
    for (Py_ssize_t i = 0; i < n; ++i) {
      if (threaded[i]) continue;
      int32_t* dst = (int32_t*)bufs[i].buf;
      if (walk(objs[i], shapes[i], nds[i], &dst) < 0) {
        /* tell the caller which tensor it was: the exception keeps its type and message, plus the index as an attribute-free note */
        PyObject *et, *ev, *tb;
        PyErr_Fetch(&et, &ev, &tb);
        PyErr_NormalizeException(&et, &ev, &tb);
        if (ev) { PyObject* idx = PyLong_FromSsize_t(i); if (idx) { PyObject_SetAttrString(ev, "tensor_index", idx); Py_DECREF(idx); } }
        PyErr_Restore(et, ev, tb);
        goto done;
      }
    }
  }
  res = Py_None;
  Py_INCREF(res);
done:
  for (Py_ssize_t i = 0; i < got; ++i) PyBuffer_Release(&bufs[i]);
  Py_DECREF(fast);
  return res;
}

static PyMethodDef methods[] = {{"pack", pack, METH_VARARGS, "pack(obj, out, shape, nthreads=1): nested lists -> int32 buffer"},
                                {"pack_many", pack_many, METH_VARARGS, "pack_many([(obj, out, shape), ...], nthreads=1)"},
                                {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_listpack", "nested feed lists -> int32", -1, methods};
PyMODINIT_FUNC PyInit__listpack(void) {
  PyObject* m = PyModule_Create(&moddef);
  if (m) PyModule_AddIntConstant(m, "NOGIL_THREADS", LISTPACK_NOGIL);
  return m;
}
