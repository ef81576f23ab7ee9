/* _pack -- pack a Python list of str into one code-unit buffer + offsets (CPython C API).
 *
 * Host-side glue of the string upload (what pfz_strings_upload consumes): 1-byte code units when every
 * code point is <= 0xFF (CPython's 1-byte "kind" is exactly Latin-1), UTF-32 otherwise.  Same result as
 * the pure-Python polyfuzz_amd._lib.pack_strings fallback, about 5x faster on 100k company names
 * (no intermediate joined str, no per-string Python call).  Replaces nothing in the reference: its
 * matchers hand Python lists straight to sklearn / rapidfuzz (reference models/_base.py:13-16).
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* pack(list[str] [, n_threads [, obj_addr]]) -> (code units: bytes, offsets int64[n+1]: bytes, bytes per code unit)
 * obj_addr != 0: the data address of a FRESH np.empty(n, object) array that becomes the From column of the result frame
 * (slot i = a new reference to string i) in the same walk -- a separate pass over 100 000 strings whose headers have left
 * this core's cache costs 0.9 ms of a 4 ms TFIDF.match().
 *
 * Two passes over the strings -- lengths / widest kind, then the copy of the code units -- each a walk over n
 * scattered PyUnicode objects (a cache miss per string).  Above ~16k strings both passes are split over n_threads
 * pthreads: the workers only READ immutable str objects (length, kind, data) while the calling thread keeps the GIL
 * and waits, so nothing can change under them; they make no Python API call.
 */
#include <pthread.h>

/* Reference counts are only ever edited directly where that is exact: CPython < 3.12 (every object mortal, ob_refcnt a
 * plain Py_ssize_t).  From 3.12 on None and interned strings are IMMORTAL -- Py_INCREF / Py_DECREF of them are no-ops
 * and their count must not be touched -- so there the threaded fill is off (Py_INCREF under the GIL only) and the
 * references to None that np.empty(n, object) "held" need no release. */
#ifdef Py_GIL_DISABLED
#error "_pack.c relies on the GIL (free-threaded CPython is not supported)"
#endif
#if PY_VERSION_HEX >= 0x030C0000
#define PFZ_DIRECT_REFCNT 0
#else
#define PFZ_DIRECT_REFCNT 1
#endif

static void release_overwritten_none(Py_ssize_t k)
{
#if PFZ_DIRECT_REFCNT
    Py_None->ob_refcnt -= k;                   /* None stays alive: the interpreter holds references of its own */
#else
    (void)k;                                   /* immortal None: np.empty's Py_INCREF(None) was a no-op too */
#endif
}

typedef struct {
    PyObject **items;
    Py_ssize_t lo, hi;
    int64_t *off;        /* pass 1: off[i + 1] = len(i);  pass 2: reads the prefix sums */
    char *cp;
    int wide;            /* pass 1 result: a string of this range is not 1-byte;  pass 2 input: output width */
    int bad;             /* pass 1: index + 1 of the first non-str item of the range, 0 if none */
    Py_ssize_t bad_at;
    int pass;
} pack_job;

static void *pack_worker(void *arg)
{
    pack_job *j = (pack_job *)arg;
    if (j->pass == 1) {
        for (Py_ssize_t i = j->lo; i < j->hi; ++i) {
            PyObject *s = j->items[i];
            if (!PyUnicode_Check(s) || !PyUnicode_IS_READY(s)) {   /* (not ready: legacy wstr strings -- handled serially) */
                if (!j->bad) {
                    j->bad = 1;
                    j->bad_at = i;
                }
                j->off[i + 1] = 0;
                continue;
            }
            if (PyUnicode_KIND(s) != PyUnicode_1BYTE_KIND) j->wide = 1;
            j->off[i + 1] = (int64_t)PyUnicode_GET_LENGTH(s);
        }
    } else {
        for (Py_ssize_t i = j->lo; i < j->hi; ++i) {
            PyObject *s = j->items[i];
            const Py_ssize_t len = PyUnicode_GET_LENGTH(s);
            const int64_t pos = j->off[i];
            if (!j->wide) {
                memcpy(j->cp + pos, PyUnicode_1BYTE_DATA(s), (size_t)len);
            } else {
                uint32_t *dst = (uint32_t *)j->cp + pos;
                const int kind = PyUnicode_KIND(s);
                const void *data = PyUnicode_DATA(s);
                for (Py_ssize_t k = 0; k < len; ++k) dst[k] = (uint32_t)PyUnicode_READ(kind, data, k);
            }
        }
    }
    return NULL;
}

/* the From column: slot i of a fresh object array = a new reference to string i (calling thread, under the GIL) */
static void fill_from_column(PyObject **items, PyObject **objs, Py_ssize_t n)
{
    Py_ssize_t nones = 0;
    for (Py_ssize_t i = 0; i < n; ++i) {
        if (i + 24 < n) __builtin_prefetch(items[i + 24], 1, 1);      /* the reference counts are cache misses */
        if (objs[i] == Py_None) ++nones;
        Py_INCREF(items[i]);
        objs[i] = items[i];
    }
    release_overwritten_none(nones);
}

/* objs != NULL (and more than one job): ALL jobs run on worker threads and the calling thread fills the From column
 * meanwhile -- the workers only read the strings' lengths and characters, the reference counts are the calling thread's
 * alone (it holds the GIL), so the two walks over the same 100 000 objects overlap instead of following each other */
static void run_pack_jobs(pack_job *jobs, int n_jobs, PyObject **items, PyObject **objs, Py_ssize_t n)
{
    pthread_t th[16];
    int started = 0;
    const int first = (objs && n_jobs > 1) ? 0 : 1;
    for (int t = first; t < n_jobs; ++t) {
        if (pthread_create(&th[started], NULL, pack_worker, &jobs[t]) != 0) {
            for (int u = t; u < n_jobs; ++u) pack_worker(&jobs[u]);     /* threads that could not be started */
            break;
        }
        ++started;
    }
    if (first == 1) pack_worker(&jobs[0]);
    if (objs) fill_from_column(items, objs, n);
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
}

/* The common case in ONE walk on the calling thread: every string 1-byte (Latin-1) and ready.  Lengths, characters and --
 * optionally -- the From column (objs) are taken while a string's two cache lines are here; the characters go into a
 * buffer sized by a guess (32 per string) that is doubled when it runs out.  Two walks on four threads (lengths, then
 * characters) + a third for the From column cost 1.3 ms of host time per 100 000 names, most of it cache misses taken
 * three times and thread starts.  Returns NULL without an exception when the list is not of that kind: nothing is left
 * behind (objs untouched again) and the general path below takes over. */
static PyObject *pack_one_walk(PyObject **items, Py_ssize_t n, PyObject **objs)
{
    PyObject *offs = PyBytes_FromStringAndSize(NULL, (n + 1) * (Py_ssize_t)sizeof(int64_t));
    if (!offs) return NULL;
    int64_t *op = (int64_t *)PyBytes_AS_STRING(offs);
    size_t cap = (size_t)n * 32 + 4096, pos = 0;
    PyObject *chars = PyBytes_FromStringAndSize(NULL, (Py_ssize_t)cap);
    if (!chars) {
        Py_DECREF(offs);
        return NULL;         /* (MemoryError set) */
    }
    char *buf = PyBytes_AS_STRING(chars);
    op[0] = 0;
    Py_ssize_t i = 0, nones = 0;
    for (; i < n; ++i) {
        if (i + 16 < n) {
            __builtin_prefetch(items[i + 16], 1, 1);                        /* header (the reference count is written) */
            __builtin_prefetch((const char *)items[i + 16] + 64, 0, 1);     /* ... and the line behind it: the characters */
        }
        PyObject *s = items[i];
        if (!PyUnicode_Check(s) || !PyUnicode_IS_READY(s) || PyUnicode_KIND(s) != PyUnicode_1BYTE_KIND) break;
        const size_t len = (size_t)PyUnicode_GET_LENGTH(s);
        if (pos + len > cap) {
            cap = (cap + len) * 2;
            if (_PyBytes_Resize(&chars, (Py_ssize_t)cap) < 0) {      /* (chars is NULL now, MemoryError set) */
                Py_DECREF(offs);
                if (objs) {
                    for (Py_ssize_t k = 0; k < i; ++k) {
                        Py_DECREF(objs[k]);
                        objs[k] = NULL;
                    }
                    release_overwritten_none(nones);      /* (the slots are NULL now, not None: their references go back) */
                }
                return NULL;
            }
            buf = PyBytes_AS_STRING(chars);
        }
        memcpy(buf + pos, PyUnicode_1BYTE_DATA(s), len);
        pos += len;
        op[i + 1] = (int64_t)pos;
        if (objs) {
            if (objs[i] == Py_None) ++nones;
            Py_INCREF(s);
            objs[i] = s;
        }
    }
    if (i < n) {                 /* not that kind of list: undo */
        if (objs) {
            for (Py_ssize_t k = 0; k < i; ++k) {
                Py_DECREF(objs[k]);
                objs[k] = NULL;
            }
            release_overwritten_none(nones);              /* (as above: NULL slots hold no reference to None) */
        }
        Py_DECREF(chars);
        Py_DECREF(offs);
        return NULL;
    }
    release_overwritten_none(nones);
    if (_PyBytes_Resize(&chars, (Py_ssize_t)pos) < 0) {               /* (shrinks in place) */
        Py_DECREF(offs);
        return NULL;
    }
    return Py_BuildValue("(NNi)", chars, offs, 1);
}

/* pack_into(strings, obj_addr, buf_addr, off_bytes, cap) -> number of characters, or None
 *
 * pack_one_walk() into memory the caller owns (the engine's PINNED staging buffer, pfz_stage_reserve): int64 offsets[n + 1] at
 * buf, the 1-byte characters at buf + off_bytes, cap bytes in all.  None -- and nothing left behind, objs untouched again -- when the
 * list is not all ready 1-byte str or does not fit: the caller takes pack() and the copying upload. */
static PyObject *pack_into(PyObject *self, PyObject *args)
{
    (void)self;
    PyObject *arg;
    unsigned long long obj_addr = 0, buf_addr = 0;
    Py_ssize_t off_bytes = 0, cap = 0;
    if (!PyArg_ParseTuple(args, "OKKnn", &arg, &obj_addr, &buf_addr, &off_bytes, &cap)) return NULL;
    PyObject *seq = PySequence_Fast(arg, "pack_into() expects a sequence of str");
    if (!seq) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    PyObject **items = PySequence_Fast_ITEMS(seq);
    PyObject **objs = (PyObject **)(uintptr_t)obj_addr;
    if (!buf_addr || off_bytes < (n + 1) * (Py_ssize_t)sizeof(int64_t) || cap < off_bytes) {
        Py_DECREF(seq);
        PyErr_SetString(PyExc_ValueError, "pack_into(): the buffer does not hold the offsets");
        return NULL;
    }
    if (objs)
        for (Py_ssize_t i = 0; i < n; ++i)
            if (objs[i] != NULL && objs[i] != Py_None) {
                Py_DECREF(seq);
                PyErr_SetString(PyExc_ValueError, "pack_into(): the object array must be a fresh np.empty array");
                return NULL;
            }
    int64_t *op = (int64_t *)(uintptr_t)buf_addr;
    char *buf = (char *)(uintptr_t)buf_addr + off_bytes;
    const size_t room = (size_t)(cap - off_bytes);
    size_t pos = 0;
    op[0] = 0;
    Py_ssize_t i = 0, nones = 0;
    for (; i < n; ++i) {
        if (i + 16 < n) {
            __builtin_prefetch(items[i + 16], 1, 1);
            __builtin_prefetch((const char *)items[i + 16] + 64, 0, 1);
        }
        PyObject *s = items[i];
        if (!PyUnicode_Check(s) || !PyUnicode_IS_READY(s) || PyUnicode_KIND(s) != PyUnicode_1BYTE_KIND) break;
        const size_t len = (size_t)PyUnicode_GET_LENGTH(s);
        if (pos + len > room) break;
        memcpy(buf + pos, PyUnicode_1BYTE_DATA(s), len);
        pos += len;
        op[i + 1] = (int64_t)pos;
        if (objs) {
            if (objs[i] == Py_None) ++nones;
            Py_INCREF(s);
            objs[i] = s;
        }
    }
    if (i < n) {                 /* not that kind of list, or too long: undo */
        if (objs) {
            for (Py_ssize_t k = 0; k < i; ++k) {
                Py_DECREF(objs[k]);
                objs[k] = NULL;
            }
            release_overwritten_none(nones);
        }
        Py_DECREF(seq);
        Py_RETURN_NONE;
    }
    release_overwritten_none(nones);
    Py_DECREF(seq);
    return PyLong_FromSize_t(pos);
}

static PyObject *pack(PyObject *self, PyObject *args)
{
    (void)self;
    PyObject *arg;
    int n_threads = 1;
    unsigned long long obj_addr = 0;
    if (!PyArg_ParseTuple(args, "O|iK", &arg, &n_threads, &obj_addr)) return NULL;
    PyObject *seq = PySequence_Fast(arg, "pack() expects a sequence of str");
    if (!seq) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    PyObject **items = PySequence_Fast_ITEMS(seq);
    PyObject **objs_arg = (PyObject **)(uintptr_t)obj_addr;
    if (objs_arg)
        for (Py_ssize_t i = 0; i < n; ++i)
            if (objs_arg[i] != NULL && objs_arg[i] != Py_None) {
                Py_DECREF(seq);
                PyErr_SetString(PyExc_ValueError, "pack(): the object array must be a fresh np.empty array");
                return NULL;
            }
    if (n_threads >= 0) {        /* (n_threads < 0: the general path only -- tests) */
        PyObject *fast = pack_one_walk(items, n, objs_arg);
        if (fast || PyErr_Occurred()) {
            Py_DECREF(seq);
            return fast;
        }
    } else {
        n_threads = -n_threads;
    }
    PyObject *offs = PyBytes_FromStringAndSize(NULL, (n + 1) * (Py_ssize_t)sizeof(int64_t));
    if (!offs) {
        Py_DECREF(seq);
        return NULL;
    }
    int64_t *op = (int64_t *)PyBytes_AS_STRING(offs);
    op[0] = 0;
    int n_jobs = n >= 16384 ? n_threads : 1;
    if (n_jobs < 1) n_jobs = 1;
    if (n_jobs > 16) n_jobs = 16;
    pack_job jobs[16];
    for (int t = 0; t < n_jobs; ++t) {
        jobs[t].items = items;
        jobs[t].lo = n * t / n_jobs;
        jobs[t].hi = n * (t + 1) / n_jobs;
        jobs[t].off = op;
        jobs[t].cp = NULL;
        jobs[t].wide = 0;
        jobs[t].bad = 0;
        jobs[t].bad_at = 0;
        jobs[t].pass = 1;
    }
    run_pack_jobs(jobs, n_jobs, items, NULL, n);
    int wide = 0;
    for (int t = 0; t < n_jobs; ++t) {
        if (jobs[t].bad) {      /* a non-str item, or a string that is not in canonical form yet */
            PyObject *s = items[jobs[t].bad_at];
            if (PyUnicode_Check(s)) {          /* make every string ready (serial, needs the GIL) and start over */
                for (Py_ssize_t i = 0; i < n; ++i)
                    if (PyUnicode_Check(items[i]) && PyUnicode_READY(items[i]) < 0) {
                        Py_DECREF(offs);
                        Py_DECREF(seq);
                        return NULL;
                    }
                Py_DECREF(offs);
                PyObject *again = pack(self, args);
                Py_DECREF(seq);
                return again;
            }
            PyErr_Format(PyExc_TypeError, "pack(): item %zd is %s, not str", jobs[t].bad_at, Py_TYPE(s)->tp_name);
            Py_DECREF(offs);
            Py_DECREF(seq);
            return NULL;
        }
        wide |= jobs[t].wide;
    }
    for (Py_ssize_t i = 0; i < n; ++i) op[i + 1] += op[i];
    const Py_ssize_t total = (Py_ssize_t)op[n];
    const int width = wide ? 4 : 1;
    PyObject *chars = PyBytes_FromStringAndSize(NULL, total * width);
    if (!chars) {
        Py_DECREF(offs);
        Py_DECREF(seq);
        return NULL;
    }
    PyObject **objs = (PyObject **)(uintptr_t)obj_addr;
    if (objs)
        for (Py_ssize_t i = 0; i < n; ++i)
            if (objs[i] != NULL && objs[i] != Py_None) {
                Py_DECREF(chars);
                Py_DECREF(offs);
                Py_DECREF(seq);
                PyErr_SetString(PyExc_ValueError, "pack(): the object array must be a fresh np.empty array");
                return NULL;
            }
    for (int t = 0; t < n_jobs; ++t) {
        jobs[t].cp = PyBytes_AS_STRING(chars);
        jobs[t].wide = wide;
        jobs[t].pass = 2;
    }
    run_pack_jobs(jobs, n_jobs, items, objs, n);
    Py_DECREF(seq);
    return Py_BuildValue("(NNi)", chars, offs, width);
}

/* fill_columns(names, idx_addr, val_addr, n, top_n, obj_addrs, sim_addrs, n_threads)
 *
 * The (To_r, Similarity_r) column pairs of the result frame (reference polyfuzz/models/_utils.py:104-125),
 * r = 0..top_n-1, from the engine's int32 idx[n][top_n] / fp32 val[n][top_n] result arrays:
 *   sim_r[i] = round(float64(val[i][r]), 3)            -- numpy's round: rint(x * 1000) / 1000
 *   if sim_r[i] < 0.001 or idx[i][r] is not a row of `names`:  sim_r[i] = 0.0, obj_r[i] = None
 *   else                                                        obj_r[i] = names[idx[i][r]]
 * obj_addrs / sim_addrs: tuples of the data addresses of top_n FRESH numpy object arrays (every slot NULL
 * or None, as np.empty(n, object) leaves them) and top_n float64 arrays, n elements each.
 *
 * The gathers are random reads of PyObject headers -- one cache miss per name -- so the loop prefetches a
 * few rows ahead, and above ~64k entries the (column, row-chunk) tasks are spread over n_threads pthreads.
 * The workers make no Python API call: they bump reference counts with atomic adds and store into disjoint
 * slots, while the calling thread keeps the GIL and waits, so no other reference-count update can race.
 */
typedef struct {
    PyObject **items;
    Py_ssize_t n_names, n, top_n;
    const int32_t *idx;
    const float *val;
    PyObject ***obj;
    double **sim;
    Py_ssize_t chunk, n_chunks;
    long next_task;   /* atomic */
} fill_job;

static void fill_range(const fill_job *job, Py_ssize_t r, Py_ssize_t lo, Py_ssize_t hi, int atomic)
{
    enum { AHEAD = 12 };
    const Py_ssize_t stride = job->top_n;
    const int32_t *idx = job->idx + r;
    const float *val = job->val + r;
    PyObject **obj = job->obj[r];
    double *sim = job->sim[r];
    PyObject **items = job->items;
    const Py_ssize_t n_names = job->n_names;
    for (Py_ssize_t i = lo; i < hi; ++i) {
        if (i + AHEAD < hi) {
            const int32_t jn = idx[(i + AHEAD) * stride];
            if (jn >= 0 && jn < n_names) __builtin_prefetch(items[jn], 1, 1);
        }
        const int32_t j = idx[i * stride];
        double s = rint((double)val[i * stride] * 1000.0) / 1000.0;
        PyObject *o = Py_None;
        if (s < 0.001 || j < 0 || j >= n_names) s = 0.0;
        else o = items[j];
        sim[i] = s;
#if PFZ_DIRECT_REFCNT
        if (atomic) __atomic_fetch_add(&o->ob_refcnt, 1, __ATOMIC_RELAXED);
        else Py_INCREF(o);
#else
        (void)atomic;
        Py_INCREF(o);
#endif
        obj[i] = o;
    }
}

static void *fill_worker(void *arg)
{
    fill_job *job = (fill_job *)arg;
    const long n_tasks = (long)(job->n_chunks * job->top_n);
    for (;;) {
        const long t = __atomic_fetch_add(&job->next_task, 1, __ATOMIC_RELAXED);
        if (t >= n_tasks) break;
        const Py_ssize_t r = t / job->n_chunks, c = t % job->n_chunks;
        const Py_ssize_t lo = c * job->chunk, hi = lo + job->chunk < job->n ? lo + job->chunk : job->n;
        fill_range(job, r, lo, hi, 1);
    }
    return NULL;
}

static PyObject *fill_columns(PyObject *self, PyObject *args)
{
    (void)self;
    PyObject *names, *obj_addrs, *sim_addrs;
    unsigned long long idx_addr, val_addr;
    Py_ssize_t n, top_n;
    int n_threads;
    if (!PyArg_ParseTuple(args, "OKKnnO!O!i", &names, &idx_addr, &val_addr, &n, &top_n, &PyTuple_Type, &obj_addrs,
                          &PyTuple_Type, &sim_addrs, &n_threads))
        return NULL;
    if (n < 0 || top_n < 1 || top_n > (1 << 24) || PyTuple_GET_SIZE(obj_addrs) != top_n || PyTuple_GET_SIZE(sim_addrs) != top_n) {
        PyErr_SetString(PyExc_ValueError, "fill_columns(): need top_n object and top_n float64 column addresses");
        return NULL;
    }
    PyObject *seq = PySequence_Fast(names, "fill_columns() expects a sequence of names");
    if (!seq) return NULL;
    /* (1024 column pairs on the stack; a deeper top_n -- the device side has no limit -- on the heap) */
    PyObject **obj_stack[1024];
    double *sim_stack[1024];
    PyObject ***obj = obj_stack;
    double **sim = sim_stack;
    void *heap = NULL;
    if (top_n > 1024) {
        heap = malloc((size_t)top_n * (sizeof(PyObject **) + sizeof(double *)));
        if (!heap) {
            Py_DECREF(seq);
            return PyErr_NoMemory();
        }
        obj = (PyObject ***)heap;
        sim = (double **)((char *)heap + (size_t)top_n * sizeof(PyObject **));
    }
    Py_ssize_t old_none = 0;
    for (Py_ssize_t r = 0; r < top_n; ++r) {
        obj[r] = (PyObject **)(uintptr_t)PyLong_AsUnsignedLongLong(PyTuple_GET_ITEM(obj_addrs, r));
        sim[r] = (double *)(uintptr_t)PyLong_AsUnsignedLongLong(PyTuple_GET_ITEM(sim_addrs, r));
        if (PyErr_Occurred()) {
            Py_DECREF(seq);
            free(heap);
            return NULL;
        }
        for (Py_ssize_t i = 0; i < n; ++i) {
            if (obj[r][i] == Py_None) ++old_none;     /* np.empty(n, object) holds n references to None */
            else if (obj[r][i] != NULL) {
                Py_DECREF(seq);
                free(heap);
                PyErr_SetString(PyExc_ValueError, "fill_columns(): the object columns must be fresh np.empty arrays");
                return NULL;
            }
        }
    }
    fill_job job;
    job.items = PySequence_Fast_ITEMS(seq);
    job.n_names = PySequence_Fast_GET_SIZE(seq);
    job.n = n;
    job.top_n = top_n;
    job.idx = (const int32_t *)(uintptr_t)idx_addr;
    job.val = (const float *)(uintptr_t)val_addr;
    job.obj = obj;
    job.sim = sim;
    job.chunk = 8192;
    job.n_chunks = (n + job.chunk - 1) / job.chunk;
    job.next_task = 0;
    const long n_tasks = (long)(job.n_chunks * top_n);
    if (n_threads > 16) n_threads = 16;
    if (n_threads > n_tasks) n_threads = (int)n_tasks;
#if !PFZ_DIRECT_REFCNT
    n_threads = 1;                             /* no atomic reference counts next to immortal objects */
#endif
    int started = 0;
    pthread_t th[16];
    if (n * top_n >= 65536 && n_threads > 1) {
        for (; started < n_threads - 1; ++started)
            if (pthread_create(&th[started], NULL, fill_worker, &job) != 0) break;
    }
    if (started > 0) {
        fill_worker(&job);                     /* the calling thread works too (atomic adds, like the others) */
        for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    } else {
        for (Py_ssize_t r = 0; r < top_n; ++r) fill_range(&job, r, 0, n, 0);
    }
    release_overwritten_none(old_none);        /* the references the overwritten slots held */
    Py_DECREF(seq);
    free(heap);
    Py_RETURN_NONE;
}

/* fill_objects(seq, obj_addr, n): the From column -- slot i of a fresh np.empty(n, object) array gets a new
 * reference to seq[i].  (numpy's `arr[:] = list` walks the list through the generic sequence protocol: 1.5 - 3 ms
 * for 100 000 strings, most of a match's host time that the device cannot hide.) */
static PyObject *fill_objects(PyObject *self, PyObject *args)
{
    (void)self;
    PyObject *src;
    unsigned long long addr;
    Py_ssize_t n;
    if (!PyArg_ParseTuple(args, "OKn", &src, &addr, &n)) return NULL;
    PyObject *seq = PySequence_Fast(src, "fill_objects() expects a sequence");
    if (!seq) return NULL;
    if (n < 0 || PySequence_Fast_GET_SIZE(seq) != n) {
        Py_DECREF(seq);
        PyErr_SetString(PyExc_ValueError, "fill_objects(): the array and the sequence differ in length");
        return NULL;
    }
    PyObject **dst = (PyObject **)(uintptr_t)addr;
    PyObject **items = PySequence_Fast_ITEMS(seq);
    Py_ssize_t old_none = 0;
    for (Py_ssize_t i = 0; i < n; ++i) {
        if (dst[i] == Py_None) ++old_none;            /* np.empty(n, object) holds n references to None */
        else if (dst[i] != NULL) {
            Py_DECREF(seq);
            PyErr_SetString(PyExc_ValueError, "fill_objects(): the object array must be a fresh np.empty array");
            return NULL;
        }
    }
    for (Py_ssize_t i = 0; i < n; ++i) {
        if (i + 24 < n) __builtin_prefetch(items[i + 24], 1, 1);      /* the reference counts are cache misses */
        Py_INCREF(items[i]);
        dst[i] = items[i];
    }
    release_overwritten_none(old_none);
    Py_DECREF(seq);
    Py_RETURN_NONE;
}

/* gather_objects(names, idx_addr, n, obj_addr, keep_addr): the To column of the edit-distance matchers -- slot i of a fresh
 * np.empty(n, object) array gets a new reference to names[idx[i]] (int32 idx), or None when idx[i] is not a row of
 * `names` or keep[i] (uint8 mask, address 0 = all kept) is zero.  Replaces a numpy fancy-index over an object pool of
 * the whole to-list (0.7 ms for 20 000 x 20 000 titles, most of what was left of EditDistance.match on the host). */
static PyObject *gather_objects(PyObject *self, PyObject *args)
{
    (void)self;
    PyObject *src;
    unsigned long long idx_addr, obj_addr, keep_addr;
    Py_ssize_t n;
    if (!PyArg_ParseTuple(args, "OKnKK", &src, &idx_addr, &n, &obj_addr, &keep_addr)) return NULL;
    PyObject *seq = PySequence_Fast(src, "gather_objects() expects a sequence");
    if (!seq) return NULL;
    const Py_ssize_t n_names = PySequence_Fast_GET_SIZE(seq);
    PyObject **items = PySequence_Fast_ITEMS(seq);
    PyObject **dst = (PyObject **)(uintptr_t)obj_addr;
    const int32_t *idx = (const int32_t *)(uintptr_t)idx_addr;
    const uint8_t *keep = (const uint8_t *)(uintptr_t)keep_addr;
    Py_ssize_t old_none = 0;
    for (Py_ssize_t i = 0; i < n; ++i) {
        if (dst[i] == Py_None) ++old_none;            /* np.empty(n, object) holds n references to None */
        else if (dst[i] != NULL) {
            Py_DECREF(seq);
            PyErr_SetString(PyExc_ValueError, "gather_objects(): the object array must be a fresh np.empty array");
            return NULL;
        }
    }
    enum { AHEAD = 12 };
    for (Py_ssize_t i = 0; i < n; ++i) {
        if (i + AHEAD < n) {
            const int32_t jn = idx[i + AHEAD];
            if (jn >= 0 && jn < n_names) __builtin_prefetch(items[jn], 1, 1);
        }
        const int32_t j = idx[i];
        PyObject *o = (j >= 0 && j < n_names && (!keep || keep[i])) ? items[j] : Py_None;
        Py_INCREF(o);
        dst[i] = o;
    }
    release_overwritten_none(old_none);
    Py_DECREF(seq);
    Py_RETURN_NONE;
}

/* linkage(from_ids, to_ids, n_strings) -> (cluster: bytes int32[n_strings], order: bytes int32[n_mapped])
 *
 * The order-dependent greedy assignment of reference polyfuzz/linkage.py:28-45 on integer string ids
 * (the rows of `matches` that passed the similarity filter, in frame order):
 *     if not mapping.get(From):            # unmapped -- or mapped to cluster 0, which is falsy
 *         if not mapping.get(To):  mapping[To] = mapping[From] = cluster_id; cluster_id += 1
 *         else:                    mapping[From] = mapping[To]
 * cluster[x] = final cluster id of string x (-1: never mapped); order = the string ids in the order they
 * first entered the mapping (Python dict insertion order, which the reference's `clusters` lists and
 * `cluster_name_map` representatives follow).  from_ids / to_ids: int32 buffers of equal length.
 */
static PyObject *linkage(PyObject *self, PyObject *args)
{
    (void)self;
    Py_buffer fb, tb;
    Py_ssize_t n_strings;
    if (!PyArg_ParseTuple(args, "y*y*n", &fb, &tb, &n_strings)) return NULL;
    PyObject *out = NULL, *cl = NULL, *od = NULL;
    const Py_ssize_t m = fb.len / (Py_ssize_t)sizeof(int32_t);
    if (fb.len != tb.len || fb.len % (Py_ssize_t)sizeof(int32_t) || n_strings < 0) {
        PyErr_SetString(PyExc_ValueError, "linkage(): from_ids and to_ids must be int32 buffers of equal length");
        goto done;
    }
    cl = PyBytes_FromStringAndSize(NULL, n_strings * (Py_ssize_t)sizeof(int32_t));
    od = PyBytes_FromStringAndSize(NULL, n_strings * (Py_ssize_t)sizeof(int32_t));
    if (!cl || !od) goto done;
    {
        const int32_t *f = (const int32_t *)fb.buf, *t = (const int32_t *)tb.buf;
        int32_t *map = (int32_t *)PyBytes_AS_STRING(cl), *order = (int32_t *)PyBytes_AS_STRING(od);
        Py_ssize_t n_order = 0;
        int32_t next = 0;
        for (Py_ssize_t i = 0; i < n_strings; ++i) map[i] = -1;
        for (Py_ssize_t r = 0; r < m; ++r) {
            const int32_t a = f[r], b = t[r];
            if (a < 0 || a >= n_strings || b < 0 || b >= n_strings) {
                PyErr_SetString(PyExc_ValueError, "linkage(): string id out of range");
                goto done;
            }
            if (map[a] > 0) continue;                       /* From already mapped (0 is falsy) */
            if (map[b] <= 0) {                              /* To unmapped too: a new cluster */
                if (map[b] < 0) order[n_order++] = b;
                map[b] = next;
                if (map[a] < 0) order[n_order++] = a;       /* (a == b: already entered) */
                map[a] = next;
                ++next;
            } else {
                if (map[a] < 0) order[n_order++] = a;
                map[a] = map[b];
            }
        }
        if (_PyBytes_Resize(&od, n_order * (Py_ssize_t)sizeof(int32_t)) < 0) goto done;
    }
    out = Py_BuildValue("(OO)", cl, od);
done:
    Py_XDECREF(cl);
    Py_XDECREF(od);
    PyBuffer_Release(&fb);
    PyBuffer_Release(&tb);
    return out;
}

static PyMethodDef methods[] = {
    {"linkage", linkage, METH_VARARGS, "linkage(from_ids, to_ids, n_strings) -> (cluster int32[n], order int32[k]) as bytes"},
    {"pack", pack, METH_VARARGS, "pack(list[str] [, n_threads]) -> (code units: bytes, offsets int64[n+1]: bytes, bytes per code unit)"},
    {"gather_objects", gather_objects, METH_VARARGS, "gather_objects(names, idx_addr, n, obj_addr, keep_addr): obj[i] <- names[idx[i]] or None"},
    {"pack_into", pack_into, METH_VARARGS, "pack_into(strings, obj_addr, buf_addr, off_bytes, cap): offsets + 1-byte characters into the caller's buffer -> characters, or None"},
    {"fill_objects", fill_objects, METH_VARARGS, "fill_objects(seq, obj_addr, n): a fresh object array <- new references to seq[i]"},
    {"fill_columns", fill_columns, METH_VARARGS,
     "fill_columns(names, idx_addr, val_addr, n, top_n, obj_addrs, sim_addrs, n_threads): the (To, Similarity) column pairs"},
    {NULL, NULL, 0, NULL},
};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_pack", NULL, -1, methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__pack(void) { return PyModule_Create(&module); }
