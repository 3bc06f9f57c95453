/*
 * ref_seam.c -- TEST / MEASUREMENT INFRASTRUCTURE ONLY (bench.py's cpu_baseline.seam).
 *
 * Times the reference's C core without its Python marshalling: N pthreads call the reference's own
 * internal seam `void encode(struct EncodeTask*)` (reference include/hutoken/core.h:11, src/core.c:339-511)
 * once per document, exactly as its worker does (src/lib.c:48-57), on NUL-terminated copies of the documents
 * of a packed batch.  The functions are looked up in the compiled reference (oracle/_ref/_hutoken*.so, built
 * by oracle/Makefile from the sources under /root/reference); nothing of the reference is compiled in here.
 * The two structures are restated from include/hutoken/taskqueue.h:40-45 and include/hutoken/vector.h:6-10.
 * Loaded by ctypes inside a Python process (the reference's module needs libpython's symbols).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

struct IntVector { int* data; size_t size; size_t capacity; };                       /* vector.h:6-10 */
struct EncodeTask { char* text; void* ctx; struct IntVector* tokens; char* error_msg; }; /* taskqueue.h:40-45 */

typedef void (*encode_fn)(struct EncodeTask*);
typedef void (*vinit_fn)(struct IntVector*, size_t);      /* vector.c:10 */
typedef void (*vfree_fn)(struct IntVector*);

struct seam_job {
    encode_fn enc; vinit_fn vinit; vfree_fn vfree; void* ctx;
    char* text; const int64_t* toff; /* NUL-terminated copies and their offsets */
    int64_t n_docs; int64_t* next; pthread_mutex_t* mu;
    int64_t n_ids; uint64_t xsum;
};

static void* seam_worker(void* a) {
    struct seam_job* j = a;
    for (;;) {
        pthread_mutex_lock(j->mu);              /* one document per grab, like taskqueue.c:16-35 */
        const int64_t d = (*j->next)++;
        pthread_mutex_unlock(j->mu);
        if (d >= j->n_docs) break;
        struct IntVector v;
        j->vinit(&v, 256);                       /* lib.c:774 */
        if (!v.data) break;
        struct EncodeTask t = {j->text + j->toff[d], j->ctx, &v, NULL};
        j->enc(&t);
        j->n_ids += (int64_t)v.size;
        for (size_t i = 0; i < v.size; i++) j->xsum = j->xsum * 1099511628211ull + (uint64_t)(uint32_t)v.data[i];
        j->vfree(&v);
    }
    return NULL;
}

/* -> 0 on success; *seconds = wall time of the threaded encode, *n_ids = total ids */
int ref_seam_batch(const char* so_path, const uint8_t* bytes, const int64_t* offsets, int64_t n_docs, int threads,
                   int64_t* n_ids, double* seconds) {
    void* h = dlopen(so_path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) return 1;
    encode_fn enc = (encode_fn)dlsym(h, "encode");
    vinit_fn vinit = (vinit_fn)dlsym(h, "vector_init");
    vfree_fn vfree = (vfree_fn)dlsym(h, "vector_free");
    void** pctx = (void**)dlsym(h, "global_encode_context");
    if (!enc || !vinit || !vfree || !pctx || !*pctx) return 2;
    if (threads < 1) threads = 1;
    if (threads > 1024) threads = 1024;
    const int64_t nb = offsets[n_docs];
    char* text = malloc((size_t)(nb + n_docs + 1));
    int64_t* toff = malloc(sizeof(int64_t) * (size_t)(n_docs + 1));
    pthread_t* th = malloc(sizeof(pthread_t) * (size_t)threads);
    struct seam_job* jobs = calloc((size_t)threads, sizeof *jobs);
    if (!text || !toff || !th || !jobs) return 3;
    int64_t at = 0;
    for (int64_t d = 0; d < n_docs; d++) {
        const int64_t len = offsets[d + 1] - offsets[d];
        toff[d] = at;
        memcpy(text + at, bytes + offsets[d], (size_t)len);
        text[at + len] = 0;
        at += len + 1;
    }
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    int64_t next = 0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; t++) {
        jobs[t] = (struct seam_job){enc, vinit, vfree, *pctx, text, toff, n_docs, &next, &mu, 0, 0};
        pthread_create(&th[t], NULL, seam_worker, &jobs[t]);
    }
    int64_t total = 0;
    for (int t = 0; t < threads; t++) {
        pthread_join(th[t], NULL);
        total += jobs[t].n_ids;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    *n_ids = total;
    free(text); free(toff); free(th); free(jobs);
    return 0;
}
