/* _pack -- pack a Python list of str into one code-unit buffer + offsets (CPython C API).
 *
 * Host-side glue of the string upload (what pfz_strings_upload consumes): 1-byte code units when every
 * code point is <= 0xFF (CPython's 1-byte "kind" is exactly Latin-1), UTF-32 otherwise.  Same result as
 * the pure-Python polyfuzz_amd._lib.pack_strings fallback, about 5x faster on 100k company names
 * (no intermediate joined str, no per-string Python call).  Replaces nothing in the reference: its
 * matchers hand Python lists straight to sklearn / rapidfuzz (reference models/_base.py:13-16).
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE          /* sched_getcpu, pthread_attr_setaffinity_np */
#endif
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdio.h>
#include <time.h>

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
 * and their count must not be touched -- so there the references to None that np.empty(n, object) "held" need no release.
 * (The only direct edit left is that release; every reference this file TAKES is a Py_INCREF by the thread that holds the GIL.) */
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

/* Where the helper threads of fill_ranges / pack_into run: on the cores that share the calling thread's L3 (its CCD on an EPYC
 * host), each on a core of its own -- never on the caller's core or its SMT sibling.  The strings' headers lie in the caller's
 * caches (it has just walked them); from another CCD, let alone the other socket, every one of them is a remote miss (measured on
 * the MI355X host, 2 x EPYC 9575F: four unpinned helpers made the frame fill 4 x SLOWER).  Read from sysfs once per CPU the caller
 * is found on; any failure = no pinning.  PFZ_HOST_PIN=0 turns it off. */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <sched.h>

static int parse_cpu_list(const char *path, cpu_set_t *set)
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    char line[4096];
    if (!fgets(line, sizeof line, f)) {
        fclose(f);
        return -1;
    }
    fclose(f);
    CPU_ZERO(set);
    for (char *p = line; *p && *p != '\n';) {
        char *end;
        long a = strtol(p, &end, 10), b = a;
        if (end == p) return -1;
        if (*end == '-') {
            p = end + 1;
            b = strtol(p, &end, 10);
            if (end == p) return -1;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) CPU_SET((int)c, set);
        p = *end == ',' ? end + 1 : end;
    }
    return 0;
}

/* the caller's L3 domain without the caller's own core, as an affinity mask (what l3_local_cpus counted the cores of) */
static cpu_set_t g_l3_mask;
static int g_l3_mask_for = -1;
static cpu_set_t g_allowed_mask;            /* the calling thread's affinity when the L3 domain was looked up */
static int g_allowed_known = 0;

static int pin_is_off(void)
{
    const char *e = getenv("PFZ_HOST_PIN");
    return e && e[0] == '0';
}

/* cpus[0..k): one CPU per physical core of the caller's L3 domain, the caller's own core left out; returns k (0: do not pin) */
static int l3_local_cpus(int *cpus, int max)
{
    static int cached_for = -1, cached_n = 0, cached[64];
    if (pin_is_off()) return 0;
    const int me = sched_getcpu();
    if (me < 0) return 0;
    if (me != cached_for) {
        cached_for = me;
        cached_n = 0;
        char path[160];
        cpu_set_t l3, allowed, sib;
        snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", me);
        if (parse_cpu_list(path, &l3) != 0 || sched_getaffinity(0, sizeof allowed, &allowed) != 0) return 0;
        snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", me);
        if (parse_cpu_list(path, &sib) != 0) CPU_ZERO(&sib);
        CPU_SET(me, &sib);
        g_allowed_mask = allowed;
        g_allowed_known = 1;
        CPU_ZERO(&g_l3_mask);
        g_l3_mask_for = -1;
        for (int c = 0; c < CPU_SETSIZE; ++c)
            if (CPU_ISSET(c, &l3) && CPU_ISSET(c, &allowed) && !CPU_ISSET(c, &sib)) {
                CPU_SET(c, &g_l3_mask);
                g_l3_mask_for = me;
            }
        for (int c = 0; c < CPU_SETSIZE && cached_n < 64; ++c) {
            if (!CPU_ISSET(c, &l3) || !CPU_ISSET(c, &allowed) || CPU_ISSET(c, &sib)) continue;
            cpu_set_t its;
            snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
            if (parse_cpu_list(path, &its) == 0) {
                int first = c;
                for (int d = 0; d < c; ++d)
                    if (CPU_ISSET(d, &its)) {
                        first = d;
                        break;
                    }
                if (first != c && CPU_ISSET(first, &l3) && CPU_ISSET(first, &allowed)) continue;      /* (its core is taken by a lower sibling) */
            }
            cached[cached_n++] = c;
        }
    }
    int k = 0;
    for (; k < cached_n && k < max; ++k) cpus[k] = cached[k];
    return k;
}

/* A crew: the helper threads of ONE call (pack_into on threads, fill_ranges), taken from a POOL of parked threads.
 * - The pool's threads are confined to the caller's L3 domain minus the caller's own core -- to the whole domain, not to a core
 *   each: the scheduler picks idle cores among them (a helper nailed to one core waits whenever a neighbour on the host has that
 *   core).  They are created on first use, detached, and sleep on a futex between calls: a call wakes as many as it wants with ONE
 *   system call (started per call -- as a chain, each helper starting the next -- the eighth helper of the string packer was there
 *   after 0.2 ms of a 0.3-ms job).  The caller's L3 domain is looked up again when the caller is found on another CPU, and the pool
 *   re-confined; a forked child starts a pool of its own.
 * - A woken thread JOINS the crew that is current, under the pool's lock, and takes a reference on it there: the job lives on the
 *   heap behind the crew's header and is freed by whoever lets go of it last.  The caller closes the crew (nobody can join any more)
 *   when the job's work is done.
 * - NOBODY waits for a helper as such: the work is drawn from counters and the phases end when the work of the phase is done,
 *   whoever did it.  A helper the scheduler has not run yet (placed behind a spinning thread, on a core that sleeps) does not hold
 *   the call up -- with a barrier over the threads one call in ten took 6 - 13 ms instead of 3 on a freshly started box; it finds
 *   the crew closed, or nothing left to draw, and goes back to sleep. */
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

typedef struct crew {
    int refs;                               /* atomic: the caller + every helper that has joined */
    int planned;                            /* helpers wanted */
    int joined;                             /* helpers that have joined (under the pool's lock): their tids are 1 .. joined */
    void (*run)(void *job, int tid);        /* (the caller is tid 0 and does its part itself) */
    _Alignas(64) char job[];
} crew;

static struct {
    pthread_mutex_t lock;
    int n;                                  /* threads alive */
    pid_t pid;                              /* the process they belong to (a forked child has none of them) */
    int mask_for;                           /* what the pool is confined to: g_l3_mask_for's value then, -1 = not confined */
    int mask_gen;                           /* bumped when `mask` changes: a thread that sees another value confines itself anew */
    cpu_set_t mask;
    uint32_t gen;                           /* futex word: bumped by every dispatch */
    crew *current;                          /* the crew helpers may join, or NULL */
} g_pool = {.lock = PTHREAD_MUTEX_INITIALIZER, .mask_for = -1};

static crew *crew_new(size_t job_bytes, int helpers, void (*run)(void *, int))
{
    crew *c = (crew *)calloc(1, sizeof(crew) + job_bytes);
    if (!c) return NULL;
    if (helpers > 15) helpers = 15;
    if (helpers < 0) helpers = 0;
    c->planned = helpers;
    c->refs = 1;
    c->run = run;
    return c;
}

static void crew_release(crew *c, int n)
{
    if (n > 0 && __atomic_sub_fetch(&c->refs, n, __ATOMIC_ACQ_REL) == 0) free(c);
}

static void *pool_thread(void *arg)
{
    (void)arg;
    uint32_t seen = 0;
    int my_mask_gen = 0;
    for (;;) {
        /* sleep until a dispatch bumps the word (a bump between the load and the call makes the call return at once) */
        while (__atomic_load_n(&g_pool.gen, __ATOMIC_ACQUIRE) == seen) syscall(SYS_futex, &g_pool.gen, FUTEX_WAIT_PRIVATE, seen, NULL, NULL, 0);
        seen = __atomic_load_n(&g_pool.gen, __ATOMIC_ACQUIRE);
        crew *c = NULL;
        int tid = 0, confine_anew = 0;
        cpu_set_t mask;
        pthread_mutex_lock(&g_pool.lock);
        if (my_mask_gen != g_pool.mask_gen) {
            my_mask_gen = g_pool.mask_gen;
            mask = g_pool.mask;
            confine_anew = 1;
        }
        if (g_pool.current && g_pool.current->joined < g_pool.current->planned) {
            c = g_pool.current;
            tid = ++c->joined;
            __atomic_add_fetch(&c->refs, 1, __ATOMIC_ACQ_REL);
        }
        pthread_mutex_unlock(&g_pool.lock);
        if (confine_anew) (void)sched_setaffinity(0, sizeof mask, &mask);      /* (refused: the thread stays where it may run) */
        if (c) {
            c->run(c->job, tid);
            crew_release(c, 1);
        }
    }
    return NULL;
}

/* make `c` the current crew and wake its helpers (the pool is grown to c->planned threads and confined to the caller's L3 first;
 * threads that cannot be started are simply not there: the caller does their part) */
static void crew_dispatch(crew *c)
{
    if (c->planned <= 0) return;
    const int confine = g_l3_mask_for >= 0 && !pin_is_off();
    if (g_pool.pid != getpid()) {           /* first use, or a forked child: no thread of the pool lives here (nor whoever held its lock) */
        pthread_mutex_init(&g_pool.lock, NULL);
        pthread_mutex_lock(&g_pool.lock);
        g_pool.pid = getpid();
        g_pool.n = 0;
        g_pool.current = NULL;
        g_pool.mask_for = -1;               /* (its threads will start unconfined, with mask_gen 0 behind the pool's) */
    } else {
        pthread_mutex_lock(&g_pool.lock);
    }
    while (g_pool.n < c->planned) {
        pthread_attr_t attr;
        pthread_t th;
        if (pthread_attr_init(&attr) != 0) break;
        pthread_attr_setdetachstate(&attr, PTHREAD_CREATE_DETACHED);
        pthread_attr_setstacksize(&attr, 256 * 1024);
        const int rc = pthread_create(&th, &attr, pool_thread, NULL);
        pthread_attr_destroy(&attr);
        if (rc != 0) break;
        ++g_pool.n;
    }
    const int want_mask = confine ? g_l3_mask_for : -1;
    if (g_pool.mask_for != want_mask && (confine || g_allowed_known)) {
        g_pool.mask = confine ? g_l3_mask : g_allowed_mask;      /* (the caller's L3 domain, or everything the process may use again) */
        g_pool.mask_for = want_mask;
        ++g_pool.mask_gen;
    }
    g_pool.current = c;
    pthread_mutex_unlock(&g_pool.lock);
    __atomic_add_fetch(&g_pool.gen, 1, __ATOMIC_RELEASE);
    syscall(SYS_futex, &g_pool.gen, FUTEX_WAKE_PRIVATE, c->planned, NULL, NULL, 0);
}

/* nobody can join any more (the job's work is done) */
static void crew_close(crew *c)
{
    pthread_mutex_lock(&g_pool.lock);
    if (g_pool.current == c) g_pool.current = NULL;
    pthread_mutex_unlock(&g_pool.lock);
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

/* pack_into on threads (n_threads > 1, no From column, >= 16 384 strings).  The list is cut into chunks of 2 048 strings that the
 * caller and its crew DRAW (an atomic counter) --
 * walk 1: kind checks + lengths (off[i + 1] = len) and the chunk's total; when every chunk of walk 1 is DONE (a count of chunks,
 * not of threads) everybody adds up the totals in front of the chunks it draws in walk 2: the characters to their place,
 * off[i + 1] = the end.  The threads only READ the strings (the calling thread keeps the GIL and is one of them); no reference
 * count is touched -- the frame's From column is the range fill's business then (fill_ranges). */
enum { PINTO_CHUNK = 2048 };

typedef struct {
    PyObject **items;
    Py_ssize_t n;
    int64_t *off;
    char *buf;
    size_t room;
    int n_chunks;
    int bad;                           /* atomic */
    int next1, done1, next2, done2;    /* atomic: the chunk draws of the two walks and the chunks finished */
    size_t total[];                    /* per chunk */
} pinto_job;

static void pinto_run(void *arg, int tid)
{
    (void)tid;
    pinto_job *job = (pinto_job *)arg;
    PyObject **items = job->items;
    int64_t *op = job->off;
    const int n_chunks = job->n_chunks;
    for (;;) {
        const int c = __atomic_fetch_add(&job->next1, 1, __ATOMIC_RELAXED);
        if (c >= n_chunks) break;
        const Py_ssize_t lo = (Py_ssize_t)c * PINTO_CHUNK, hi = lo + PINTO_CHUNK < job->n ? lo + PINTO_CHUNK : job->n;
        size_t sum = 0;
        for (Py_ssize_t i = lo; i < hi; ++i) {
            if (i + 16 < hi) __builtin_prefetch(items[i + 16], 0, 1);
            PyObject *s = items[i];
            if (!PyUnicode_Check(s) || !PyUnicode_IS_READY(s) || PyUnicode_KIND(s) != PyUnicode_1BYTE_KIND) {
                __atomic_store_n(&job->bad, 1, __ATOMIC_RELAXED);
                break;
            }
            const size_t len = (size_t)PyUnicode_GET_LENGTH(s);
            op[i + 1] = (int64_t)len;
            sum += len;
        }
        job->total[c] = sum;
        __atomic_add_fetch(&job->done1, 1, __ATOMIC_RELEASE);
    }
    if (__atomic_load_n(&job->next2, __ATOMIC_RELAXED) >= n_chunks) return;        /* (a late-comer: nothing left to draw) */
    while (__atomic_load_n(&job->done1, __ATOMIC_ACQUIRE) < n_chunks) __builtin_ia32_pause();
    if (__atomic_load_n(&job->bad, __ATOMIC_RELAXED)) return;
    size_t all = 0;
    for (int c = 0; c < n_chunks; ++c) all += job->total[c];
    if (all > job->room) return;          /* (everybody computes the same sum and leaves; the caller reports it) */
    char *buf = job->buf;
    int summed = 0;                        /* chunks [0, summed) are added up in pos */
    size_t pos = 0;
    for (;;) {
        const int c = __atomic_fetch_add(&job->next2, 1, __ATOMIC_RELAXED);
        if (c >= n_chunks) break;
        for (; summed < c; ++summed) pos += job->total[summed];
        const Py_ssize_t lo = (Py_ssize_t)c * PINTO_CHUNK, hi = lo + PINTO_CHUNK < job->n ? lo + PINTO_CHUNK : job->n;
        size_t at = pos;
        for (Py_ssize_t i = lo; i < hi; ++i) {
            if (i + 16 < hi) {
                __builtin_prefetch(items[i + 16], 0, 1);
                __builtin_prefetch((const char *)items[i + 16] + 64, 0, 1);
            }
            const size_t len = (size_t)op[i + 1];
            memcpy(buf + at, PyUnicode_1BYTE_DATA(items[i]), len);
            at += len;
            op[i + 1] = (int64_t)at;
        }
        __atomic_add_fetch(&job->done2, 1, __ATOMIC_RELEASE);
    }
}

/* -> characters packed; -1: not that kind of list / does not fit (nothing left behind that matters: the caller's buffer only);
 * -2: no helper could be placed on the caller's L3 (or no memory for the job) -- the caller takes the single walk */
static Py_ssize_t pack_into_threads(PyObject **items, Py_ssize_t n, int64_t *op, char *buf, size_t room, int n_threads)
{
    int local[16];
    const int n_local = l3_local_cpus(local, 16);
    if (n_threads > 16) n_threads = 16;
    if (!pin_is_off()) {
        if (n_local == 0) return -2;
        if (n_threads > 1 + n_local) n_threads = 1 + n_local;
    }
    const int n_chunks = (int)((n + PINTO_CHUNK - 1) / PINTO_CHUNK);
    crew *c = crew_new(sizeof(pinto_job) + (size_t)n_chunks * sizeof(size_t), n_threads - 1, pinto_run);
    if (!c) return -2;
    pinto_job *job = (pinto_job *)c->job;
    job->items = items;
    job->n = n;
    job->off = op;
    job->buf = buf;
    job->room = room;
    job->n_chunks = n_chunks;
    op[0] = 0;
    crew_dispatch(c);
    pinto_run(job, 0);
    /* the call is over when the WORK is: every chunk of walk 2 done -- or walk 1 done and found wanting */
    Py_ssize_t got = -1;
    while (__atomic_load_n(&job->done1, __ATOMIC_ACQUIRE) < n_chunks) __builtin_ia32_pause();
    if (!__atomic_load_n(&job->bad, __ATOMIC_RELAXED)) {
        size_t all = 0;
        for (int k = 0; k < n_chunks; ++k) all += job->total[k];
        if (all <= room) {
            while (__atomic_load_n(&job->done2, __ATOMIC_ACQUIRE) < n_chunks) __builtin_ia32_pause();
            got = (Py_ssize_t)all;
        }
    }
    crew_close(c);
    crew_release(c, 1);
    return got;
}

/* pack_into(strings, obj_addr, buf_addr, off_bytes, cap [, n_threads]) -> number of characters, or None
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
    int n_threads = 1;
    if (!PyArg_ParseTuple(args, "OKKnn|i", &arg, &obj_addr, &buf_addr, &off_bytes, &cap, &n_threads)) return NULL;
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
    if (!objs && n_threads > 1 && n >= 16384) {
        const Py_ssize_t got = pack_into_threads(items, n, op, buf, room, n_threads);
        if (got != -2) {
            Py_DECREF(seq);
            if (got < 0) Py_RETURN_NONE;
            return PyLong_FromSsize_t(got);
        }
    }
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
 * The gathers are random reads of PyObject headers -- one cache miss per name -- so the loop prefetches a few rows ahead.  On the
 * calling thread (n_threads is accepted and ignored: workers with atomic reference counts, rounds 2-5's opt-in, were slower than
 * one thread -- a big fill goes to fill_ranges' crew instead, which touches no count while it gathers).
 */
typedef struct {
    PyObject **items;
    Py_ssize_t n_names, n, top_n;
    const int32_t *idx;
    const float *val;
    PyObject ***obj;
    double **sim;
} fill_job;

static void fill_range(const fill_job *job, Py_ssize_t r, Py_ssize_t lo, Py_ssize_t hi)
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
        Py_INCREF(o);
        obj[i] = o;
    }
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
    (void)n_threads;
    for (Py_ssize_t r = 0; r < top_n; ++r) fill_range(&job, r, 0, n);
    release_overwritten_none(old_none);        /* the references the overwritten slots held */
    Py_DECREF(seq);
    free(heap);
    Py_RETURN_NONE;
}

/* fill_ranges(names, idx_addr, val_addr, top_n, obj_addrs, sim_addrs, ends, wait_addr, ctx_addr, first_slot, n_threads
 *             [, stamps_addr [, from_obj_addr]])
 *
 * The (To_r, Similarity_r) column pairs of a BIG match whose result arrives in ascending row ranges (pfz_cossim_topn_ranges with a
 * mirror: idx / val are the whole result in pinned host memory, range i = rows [ends[i-1], ends[i]) valid once
 * wait(ctx, first_slot + i) -- pfz_event_wait, a spin on a word in pinned memory -- has returned 0; wait_addr 0: everything is
 * there, any big result in ordinary memory).  Same cells as fill_columns.
 * from_obj_addr != 0: the match is a list against itself (n rows = len(names)) and the From column -- a fresh object array, slot
 * i <- a new reference to names[i] -- is filled by the calling thread before it turns to the ranges.
 *
 * Round 6: the frame's gathers were the user-level call's tail -- from the seventh of twelve ranges on the host found the ranges
 * waiting, and the call ended 0.4 - 0.8 ms behind the device.  What was measured on the way here (the MI355X host, 2 x EPYC 9575F;
 * tools/ubench/frame_fill_mt.c is the model, tools/r6_match_ab.py the call):
 *   - threads with atomic reference counts in the gathers lose (3.9 ms on four against 3.0 on one);
 *   - helper threads the scheduler places lose whatever they do: 0.44 -> 0.95 ms for the bare STORES on two threads -- the names and
 *     the columns lie in the caller's caches, from another CCD or the other socket every line is a remote miss; confined to the cores
 *     that share the caller's L3 the same stores take 0.27 / 0.16 / 0.13 ms on 2 / 4 / 8 threads (the crew above, l3_local_cpus);
 *   - every object owned by one thread by its address, each thread scanning all cells for its own: 4 x slower than one thread;
 *   - threads that count what they store, per list position, in arrays of their own, the counts added to the objects in a closing
 *     walk (atomic adds: one object may sit at several positions): the gathers keep up with the device, the walk is 0.10 ms behind
 *     the last range -- and as much when it is taken early, in front of the last ranges.
 * So the work is cut along what it touches.  **The helpers do everything that touches no reference count**: for a task (1024 rows,
 * every column) the rounding, the similarity stores, the gather of the names' pointers and the pointer stores -- then they flag the
 * task.  **The calling thread does nothing but reference counts**: it follows the flags in task order and takes one reference per
 * pointer it finds in the task's column cells (plain Py_INCREF: it holds the GIL from the first store to the last count, nobody
 * else edits a count, any CPython) -- the names' headers stay in ITS caches, where the packer's walk and the last frame's disposal
 * left them, and its work per cell is a load and an increment.  When the last range's last task is flagged, the frame is 5 000
 * increments away from done: no closing walk.
 * A helper that draws a task of a range nobody has seen final yet waits for it itself; a task nobody has drawn when the caller
 * gets to it is the caller's (no helper could be started, or none has been scheduled yet: the call does not depend on them).
 * The slots' old contents (np.empty's None or NULL) are checked and counted by whoever overwrites them.  A wait that fails stops
 * the work at that task: the tasks behind it are flagged as skipped, what was stored is counted (the columns stay consistent),
 * then it raises.
 */
typedef int (*pfz_wait_fn)(void *ctx, int32_t slot);

enum { RFILL_ROWS = 1024 };

typedef struct {
    PyObject **items;
    Py_ssize_t n_names, top_n;
    const int32_t *idx;
    const float *val;
    pfz_wait_fn wait;
    void *ctx;
    int first_slot;
    int n_ranges;
    long n_tasks;
    long next_ticket;        /* atomic */
    int error;               /* atomic: 1 = a wait failed, 2 = a slot was not fresh */
    double *stamps;          /* 2 x n_ranges (seconds, CLOCK_MONOTONIC): range seen final / its last task counted; or NULL */
    long none_old[16];       /* per thread: overwritten slots that held None */
    int64_t ends[64];
    long task_end[64];       /* task_end[i] = number of tasks of ranges 0..i */
    int ready[64];           /* range i has been seen final (atomic) */
    PyObject **obj[1024];
    double *sim[1024];
    int done[];              /* per task (atomic): 0 not yet, 1 stored, 2 skipped */
} rfill_job;

static double mono_now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* rows [lo, hi) of every column: similarities and pointers stored, no reference taken */
static void rfill_store(rfill_job *job, int tid, Py_ssize_t lo, Py_ssize_t hi)
{
    const Py_ssize_t top_n = job->top_n, n_names = job->n_names;
    PyObject *dummy = Py_None;
    PyObject **items = n_names ? job->items : &dummy;
    long none_old = 0;
    int stale = 0;
    for (Py_ssize_t r = 0; r < top_n; ++r) {
        const int32_t *idx = job->idx + r;
        const float *val = job->val + r;
        PyObject **obj = job->obj[r];
        double *sim = job->sim[r];
        for (Py_ssize_t i = lo; i < hi; ++i) {
            const int32_t j = idx[i * top_n];
            const double t = rint((double)val[i * top_n] * 1000.0);       /* numpy's round(x, 3) = rint(x * 1000) / 1000 */
            const int keep = !(t < 1.0) & (j >= 0) & (j < n_names);       /* (rint(.) / 1000 < 0.001 exactly when rint(.) < 1) */
            PyObject *o = items[keep ? j : 0];
            o = keep ? o : Py_None;
            PyObject *old = obj[i];
            none_old += old == Py_None;
            stale |= (old != NULL) & (old != Py_None);
            sim[i] = keep ? t / 1000.0 : 0.0;
            obj[i] = o;
        }
    }
    job->none_old[tid] += none_old;
    if (stale) __atomic_store_n(&job->error, 2, __ATOMIC_RELAXED);
}

/* task t, drawn by thread tid: wait for its range if nobody has seen it final, store, flag */
static void rfill_ticket(rfill_job *job, int tid, long t)
{
    int state = 2;
    if (__atomic_load_n(&job->error, __ATOMIC_RELAXED) != 1) {
        int range = 0;
        while (t >= job->task_end[range]) ++range;
        int ok = 1;
        if (job->wait && !__atomic_load_n(&job->ready[range], __ATOMIC_ACQUIRE)) {
            if (job->wait(job->ctx, job->first_slot + range) != 0) {
                __atomic_store_n(&job->error, 1, __ATOMIC_RELAXED);
                ok = 0;
            } else if (!__atomic_exchange_n(&job->ready[range], 1, __ATOMIC_ACQ_REL) && job->stamps) {
                job->stamps[2 * range] = mono_now();
            }
        }
        if (ok) {
            const Py_ssize_t row0 = range ? (Py_ssize_t)job->ends[range - 1] : 0, row1 = (Py_ssize_t)job->ends[range];
            const long k = t - (range ? job->task_end[range - 1] : 0);
            const Py_ssize_t lo = row0 + k * RFILL_ROWS, hi = lo + RFILL_ROWS < row1 ? lo + RFILL_ROWS : row1;
            rfill_store(job, tid, lo, hi);
            state = 1;
        }
    }
    __atomic_store_n(&job->done[t], state, __ATOMIC_RELEASE);
}

static void rfill_helper(void *arg, int tid)
{
    rfill_job *job = (rfill_job *)arg;
    for (;;) {
        const long t = __atomic_fetch_add(&job->next_ticket, 1, __ATOMIC_RELAXED);
        if (t >= job->n_tasks) break;
        rfill_ticket(job, tid, t);
    }
}

/* the calling thread: one reference per pointer the stored task holds; returns the cells that hold None */
static long rfill_take(rfill_job *job, long t)
{
    enum { AHEAD = 8 };
    int range = 0;
    while (t >= job->task_end[range]) ++range;
    const Py_ssize_t row0 = range ? (Py_ssize_t)job->ends[range - 1] : 0, row1 = (Py_ssize_t)job->ends[range];
    const long k = t - (range ? job->task_end[range - 1] : 0);
    const Py_ssize_t lo = row0 + k * RFILL_ROWS, hi = lo + RFILL_ROWS < row1 ? lo + RFILL_ROWS : row1;
    long nones = 0;
    for (Py_ssize_t r = 0; r < job->top_n; ++r) {
        PyObject **obj = job->obj[r];
        for (Py_ssize_t i = lo; i < hi; ++i) {
            if (i + AHEAD < hi) __builtin_prefetch(obj[i + AHEAD], 1, 1);
            PyObject *o = obj[i];
            nones += o == Py_None;
            Py_INCREF(o);                 /* (None's too: counted one by one like any other object's) */
        }
    }
    if (job->stamps && t + 1 == job->task_end[range]) job->stamps[2 * range + 1] = mono_now();
    return nones;
}

static PyObject *fill_ranges(PyObject *self, PyObject *args)
{
    (void)self;
    PyObject *names, *obj_addrs, *sim_addrs, *ends_t;
    unsigned long long idx_addr, val_addr, wait_addr, ctx_addr, stamps_addr = 0, from_obj_addr = 0;
    Py_ssize_t top_n;
    int first_slot, n_threads;
    if (!PyArg_ParseTuple(args, "OKKnO!O!O!KKii|KK", &names, &idx_addr, &val_addr, &top_n, &PyTuple_Type, &obj_addrs, &PyTuple_Type,
                          &sim_addrs, &PyTuple_Type, &ends_t, &wait_addr, &ctx_addr, &first_slot, &n_threads, &stamps_addr, &from_obj_addr))
        return NULL;
    const Py_ssize_t n_ranges = PyTuple_GET_SIZE(ends_t);
    if (top_n < 1 || top_n > 1024 || PyTuple_GET_SIZE(obj_addrs) != top_n || PyTuple_GET_SIZE(sim_addrs) != top_n || n_ranges < 1 ||
        n_ranges > 64 || !idx_addr || !val_addr) {
        PyErr_SetString(PyExc_ValueError, "fill_ranges(): need 1..1024 column pairs, 1..64 range ends and the result's addresses");
        return NULL;
    }
    PyObject *seq = PySequence_Fast(names, "fill_ranges() expects a sequence of names");
    if (!seq) return NULL;
    int64_t ends[64], prev = 0;
    long tasks = 0, task_end[64];
    for (Py_ssize_t i = 0; i < n_ranges && !PyErr_Occurred(); ++i) {
        ends[i] = (int64_t)PyLong_AsLongLong(PyTuple_GET_ITEM(ends_t, i));
        if (!PyErr_Occurred() && ends[i] <= prev) PyErr_SetString(PyExc_ValueError, "fill_ranges(): range ends must ascend from above 0");
        tasks += (long)((ends[i] - prev + RFILL_ROWS - 1) / RFILL_ROWS);
        task_end[i] = tasks;
        prev = ends[i];
    }
    if (!PyErr_Occurred() && from_obj_addr && PySequence_Fast_GET_SIZE(seq) != (Py_ssize_t)prev)
        PyErr_SetString(PyExc_ValueError, "fill_ranges(): a From column needs a list against itself (as many rows as names)");
    if (PyErr_Occurred()) {
        Py_DECREF(seq);
        return NULL;
    }
    /* helpers only where they can sit on the caller's L3 (anywhere else they cost more than they do: see above) */
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 16) n_threads = 16;
    if (n_threads > tasks) n_threads = (int)tasks;
    int local[16];
    const int n_local = n_threads > 1 ? l3_local_cpus(local, 16) : 0;
    if (n_threads > 1 + n_local && !pin_is_off()) n_threads = 1 + n_local;
    crew *c = crew_new(sizeof(rfill_job) + (size_t)tasks * sizeof(int), n_threads - 1, rfill_helper);
    if (!c) {
        Py_DECREF(seq);
        return PyErr_NoMemory();
    }
    rfill_job *job = (rfill_job *)c->job;
    for (Py_ssize_t r = 0; r < top_n; ++r) {
        job->obj[r] = (PyObject **)(uintptr_t)PyLong_AsUnsignedLongLong(PyTuple_GET_ITEM(obj_addrs, r));
        job->sim[r] = (double *)(uintptr_t)PyLong_AsUnsignedLongLong(PyTuple_GET_ITEM(sim_addrs, r));
    }
    memcpy(job->ends, ends, sizeof ends);
    memcpy(job->task_end, task_end, sizeof task_end);
    job->items = PySequence_Fast_ITEMS(seq);
    job->n_names = PySequence_Fast_GET_SIZE(seq);
    job->top_n = top_n;
    job->idx = (const int32_t *)(uintptr_t)idx_addr;
    job->val = (const float *)(uintptr_t)val_addr;
    job->n_ranges = (int)n_ranges;
    job->n_tasks = tasks;
    job->wait = (pfz_wait_fn)(uintptr_t)wait_addr;
    job->ctx = (void *)(uintptr_t)ctx_addr;
    job->first_slot = first_slot;
    job->stamps = (double *)(uintptr_t)stamps_addr;
    crew_dispatch(c);
    long none_old = 0, none_new = 0;
    int stale = 0;
    if (from_obj_addr) {          /* the From column, while the helpers wait for the first range */
        PyObject **from = (PyObject **)(uintptr_t)from_obj_addr, **items = job->items;
        for (Py_ssize_t i = 0; i < job->n_names; ++i) {
            if (i + 16 < job->n_names) __builtin_prefetch(items[i + 16], 1, 1);
            PyObject *old = from[i];
            none_old += old == Py_None;
            stale |= (old != NULL) & (old != Py_None);
            Py_INCREF(items[i]);
            from[i] = items[i];
        }
    }
    for (long t = 0; t < tasks;) {
        const int state = __atomic_load_n(&job->done[t], __ATOMIC_ACQUIRE);
        if (state) {
            if (state == 1) none_new += rfill_take(job, t);
            ++t;
            continue;
        }
        long cur = __atomic_load_n(&job->next_ticket, __ATOMIC_RELAXED);
        if (cur == t && __atomic_compare_exchange_n(&job->next_ticket, &cur, cur + 1, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
            rfill_ticket(job, 0, t);          /* (nobody has drawn it: no helper runs yet, or there is none) */
            continue;
        }
        __builtin_ia32_pause();
    }
    for (int u = 0; u < 16; ++u) none_old += job->none_old[u];
    const int error = stale ? 2 : job->error;
    crew_close(c);
    crew_release(c, 1);
    (void)none_new;                            /* (None's references were taken one by one with the others) */
    release_overwritten_none(none_old);
    Py_DECREF(seq);
    if (error == 1) {
        PyErr_SetString(PyExc_RuntimeError, "fill_ranges(): waiting for a row range of the result failed");
        return NULL;
    }
    if (error == 2) {
        PyErr_SetString(PyExc_ValueError, "fill_ranges(): the object columns must be fresh np.empty arrays");
        return NULL;
    }
    Py_RETURN_NONE;
}

/* Test seams of fill_ranges (tests/test_frame_ranges_cpu.py: no device there).  A Python callback cannot stand in for
 * pfz_event_wait -- the workers are not Python threads and the caller holds the GIL --, so the stand-in lives here:
 * test_wait(ctx = int32 flags[], slot) spins until flags[slot] != 0 and fails when it is negative; test_wait_addr() returns its
 * address; test_set_flags(addr, n, usec, value) sets flags[0..n) to value one by one, usec apart, from a detached thread. */
static int test_wait(void *ctx, int32_t slot)
{
    int32_t *flag = (int32_t *)ctx + slot, v;
    while ((v = __atomic_load_n(flag, __ATOMIC_ACQUIRE)) == 0) __builtin_ia32_pause();
    return v < 0 ? -1 : 0;
}

static PyObject *test_wait_addr(PyObject *self, PyObject *args)
{
    (void)self;
    (void)args;
    return PyLong_FromUnsignedLongLong((unsigned long long)(uintptr_t)&test_wait);
}

typedef struct {
    int32_t *flags;
    int n, usec, value;
} flag_setter;

static void *flag_setter_run(void *arg)
{
    flag_setter *f = (flag_setter *)arg;
    for (int i = 0; i < f->n; ++i) {
        struct timespec t = {0, (long)f->usec * 1000L};
        nanosleep(&t, NULL);
        __atomic_store_n(&f->flags[i], f->value, __ATOMIC_RELEASE);
    }
    free(f);
    return NULL;
}

static PyObject *test_set_flags(PyObject *self, PyObject *args)
{
    (void)self;
    unsigned long long addr;
    int n, usec, value;
    if (!PyArg_ParseTuple(args, "Kiii", &addr, &n, &usec, &value)) return NULL;
    flag_setter *f = (flag_setter *)malloc(sizeof *f);
    if (!f) return PyErr_NoMemory();
    f->flags = (int32_t *)(uintptr_t)addr;
    f->n = n;
    f->usec = usec;
    f->value = value;
    pthread_t th;
    if (pthread_create(&th, NULL, flag_setter_run, f) != 0) {
        free(f);
        PyErr_SetString(PyExc_RuntimeError, "test_set_flags(): no thread");
        return NULL;
    }
    pthread_detach(th);
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
    {"fill_ranges", fill_ranges, METH_VARARGS,
     "fill_ranges(names, idx_addr, val_addr, top_n, obj_addrs, sim_addrs, ends, wait_addr, ctx_addr, first_slot, n_threads[, stamps_addr[, from_obj_addr]]): "
     "the column pairs of a match that arrives in row ranges, on n_threads threads"},
    {"test_wait_addr", test_wait_addr, METH_NOARGS, "address of the flag-array stand-in for pfz_event_wait (tests)"},
    {"test_set_flags", test_set_flags, METH_VARARGS, "test_set_flags(addr, n, usec, value): set int32 flags one by one from a thread (tests)"},
    {"fill_objects", fill_objects, METH_VARARGS, "fill_objects(seq, obj_addr, n): a fresh object array <- new references to seq[i]"},
    {"fill_columns", fill_columns, METH_VARARGS,
     "fill_columns(names, idx_addr, val_addr, n, top_n, obj_addrs, sim_addrs, n_threads): the (To, Similarity) column pairs"},
    {NULL, NULL, 0, NULL},
};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_pack", NULL, -1, methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__pack(void) { return PyModule_Create(&module); }
