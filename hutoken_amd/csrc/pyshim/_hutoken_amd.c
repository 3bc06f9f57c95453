/*
 * _hutoken_amd.c -- CPython extension with the method table of the reference's `_hutoken` module (reference
 * src/lib.c:1128-1153) for the encode path and its decode counterpart, on top of the C ABI of include/hutoken_amd.h.
 *
 *   initialize(vocab_file_path, special_file_path, prefix=None, is_byte_encoder=False, special_token_id=-1,
 *              pattern=None, merges_file_path=None, device=-1)        lib.c:185-666, format "ss|zpizz" (+ device)
 *   encode(text) -> list[int]                                          lib.c:668-720, format "s"
 *   batch_encode(texts, num_threads=1) -> list[list[int]]              lib.c:722-874, format "O|i"
 *   decode(tokens) -> str                                              lib.c:876-951
 *   batch_decode(tokens, num_threads=1) -> list[str]                   lib.c:954-1094
 *   handle() -> int                                                    the hutk_ctx* of the module-global context
 *
 * What the reference does per DOCUMENT under the GIL (strdup of PyUnicode_AsUTF8, one EncodeTask and one IntVector
 * each, lib.c:756-777) is done here per BATCH: the texts are packed back to back (cut at their first NUL like strdup),
 * the GIL is released around hutk_encode_batch, and the result lists are built from a per-context cache of int
 * objects (one PyLong per vocabulary id, created on first use) instead of one allocation per token (lib.c:823-860).
 * Same argument errors, exception classes and messages as the reference.  No CPU path: without a GPU initialize fails.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "hutoken_amd.h"

/* Process-global like global_encode_context (lib.c:73-74).  The calls below release the GIL around the C ABI, so another
 * thread may re-initialise meanwhile: a context is destroyed only when the last call that uses it has returned (the
 * reference leaks its old contexts instead, lib.c:129-155).  Every field is touched with the GIL held. */
typedef struct ctx_box {
    hutk_ctx* ctx;
    long users;   /* calls in flight on this context */
    int retired;  /* initialize() has replaced it */
} ctx_box;
static ctx_box* g_box = NULL;
static ctx_box* box_acquire(void) {
    ctx_box* b = g_box;
    if (b) b->users++;
    return b;
}
static void box_release(ctx_box* b) {
    if (--b->users == 0 && b->retired) {
        hutk_ctx_destroy(b->ctx);
        free(b);
    }
}
static PyObject** g_int_cache = NULL; /* [g_cache_n] ints of the ids 0 .. vocabulary size - 1, NULL until first used */
static int64_t g_cache_n = 0;

static const char NOT_INIT_ENCODE[] =
    "Vocabulary is not initialized for encoding. Call 'initialize_encode' function first.";
static const char NOT_INIT_DECODE[] =
    "Vocabulary is not initialized for decoding. Call 'initialize_decode' function first.";

static PyObject* raise_code(int code) {
    PyObject* exc = PyExc_RuntimeError;
    switch (code) {
        case HUTK_E_FILE_NOT_FOUND: exc = PyExc_FileNotFoundError; break;
        case HUTK_E_VALUE: case HUTK_E_UNSUPPORTED: case HUTK_E_NUL_BYTE: case HUTK_E_INVALID_UTF8: exc = PyExc_ValueError; break;
        case HUTK_E_MEMORY: exc = PyExc_MemoryError; break;
        case HUTK_E_ARG: exc = PyExc_TypeError; break;
        default: break;
    }
    PyErr_SetString(exc, hutk_last_error());
    return NULL;
}

static void drop_cache(void) {
    if (g_int_cache) {
        for (int64_t i = 0; i < g_cache_n; i++) Py_XDECREF(g_int_cache[i]);
        free(g_int_cache);
    }
    g_int_cache = NULL;
    g_cache_n = 0;
}

static inline PyObject* id_object(int32_t id) { /* new reference */
    if (id >= 0 && (int64_t)id < g_cache_n) {
        PyObject* o = g_int_cache[id];
        if (!o) {
            o = PyLong_FromLong(id);
            if (!o) return NULL;
            g_int_cache[id] = o;
        }
        Py_INCREF(o);
        return o;
    }
    return PyLong_FromLong(id);
}

static PyObject* ids_to_list(const int32_t* ids, int64_t n) {
    PyObject* list = PyList_New((Py_ssize_t)n);
    if (!list) return NULL;
    for (int64_t i = 0; i < n; i++) {
        PyObject* o = id_object(ids[i]);
        if (!o) {
            Py_DECREF(list);
            return NULL;
        }
        PyList_SET_ITEM(list, (Py_ssize_t)i, o);
    }
    return list;
}

static PyObject* p_initialize(PyObject* self, PyObject* args, PyObject* kwargs) {
    (void)self;
    static char* kwlist[] = {"vocab_file_path", "special_file_path", "prefix", "is_byte_encoder", "special_token_id",
                             "pattern", "merges_file_path", "device", NULL};
    const char *vocab = NULL, *special = NULL, *prefix = NULL, *pattern = NULL, *merges = NULL;
    int is_byte_encoder = 0, special_token_id = -1, device = -1;
    if (!PyArg_ParseTupleAndKeywords(args, kwargs, "ss|zpizzi", kwlist, &vocab, &special, &prefix, &is_byte_encoder,
                                     &special_token_id, &pattern, &merges, &device)) {
        PyErr_SetString(PyExc_TypeError,
                        "Invalid arguments. Expected a string "
                        "(vocab_file_path), a string (special_file_path), "
                        "a string or None (prefix) a bool an"
                        "optional integer (special_token_id), "
                        " an optional string (regex_pattern) and"
                        "a string or None (merges_file_path)");
        return NULL;
    }
    hutk_ctx* ctx = NULL;
    int rc;
    Py_BEGIN_ALLOW_THREADS
    rc = hutk_ctx_create_merges(&ctx, vocab, special, prefix, is_byte_encoder, merges, device);
    if (rc == HUTK_OK && pattern) {
        rc = hutk_ctx_set_pattern(ctx, pattern);
        if (rc != HUTK_OK) {
            hutk_ctx_destroy(ctx);
            ctx = NULL;
        }
    }
    Py_END_ALLOW_THREADS
    if (rc != HUTK_OK) return raise_code(rc);
    ctx_box* box = calloc(1, sizeof *box);
    if (!box) {
        hutk_ctx_destroy(ctx);
        return PyErr_NoMemory();
    }
    box->ctx = ctx;
    ctx_box* old = g_box;
    g_box = box;
    drop_cache();
    g_cache_n = hutk_vocab_size(ctx) + 1024;  /* ids usually are 0 .. size - 1; anything else gets its own object */
    if (g_cache_n > (1 << 22)) g_cache_n = 1 << 22;
    g_int_cache = calloc((size_t)g_cache_n, sizeof(PyObject*));
    if (!g_int_cache) g_cache_n = 0;
    if (old) {  /* destroyed now, or by the last call still running on it */
        old->retired = 1;
        old->users++;
        box_release(old);
    }
    Py_RETURN_NONE;
}

static PyObject* p_handle(PyObject* self, PyObject* args) {
    (void)self;
    (void)args;
    return PyLong_FromVoidPtr(g_box ? g_box->ctx : NULL);
}

static PyObject* encode_on(hutk_ctx* ctx, PyObject* args) {
    const char* text = NULL;
    if (!PyArg_ParseTuple(args, "s", &text)) return NULL;  /* embedded NUL: ValueError, as in the reference */
    const int64_t len = (int64_t)strlen(text);
    const int64_t cap = hutk_ids_capacity(ctx, len, 1);
    int32_t* ids = malloc(sizeof(int32_t) * (size_t)(cap > 0 ? cap : 1));
    if (!ids) return PyErr_NoMemory();
    int64_t n = 0;
    int rc;
    Py_BEGIN_ALLOW_THREADS
    rc = hutk_encode(ctx, (const uint8_t*)text, len, ids, cap, &n, NULL);
    Py_END_ALLOW_THREADS
    PyObject* out = NULL;
    if (rc != HUTK_OK && rc != HUTK_E_WORD_TOO_LARGE) raise_code(rc);  /* an over-long word is not reported (lib.c:692-697) */
    else out = ids_to_list(ids, n);
    free(ids);
    return out;
}

static PyObject* batch_encode_on(hutk_ctx* ctx, PyObject* args) {
    PyObject* texts = NULL;
    int num_threads = 1;
    if (!PyArg_ParseTuple(args, "O|i", &texts, &num_threads) || !PyList_Check(texts)) {
        PyErr_SetString(PyExc_TypeError, "Invalid arguments. Expected a list of strings.");
        return NULL;
    }
    const Py_ssize_t n = PyList_GET_SIZE(texts);
    if (num_threads <= 0) {  /* no worker starts: every document stays empty (lib.c:784-791) */
        PyObject* out = PyList_New(n);
        if (!out) return NULL;
        for (Py_ssize_t i = 0; i < n; i++) {
            PyObject* e = PyList_New(0);
            if (!e) { Py_DECREF(out); return NULL; }
            PyList_SET_ITEM(out, i, e);
        }
        return out;
    }
    /* pass 1: sizes (a text ends at its first NUL, as strdup() of PyUnicode_AsUTF8 does, lib.c:770-772) */
    int64_t* offs = malloc(sizeof(int64_t) * (size_t)(n + 1));
    const char** ptrs = malloc(sizeof(char*) * (size_t)(n ? n : 1));
    if (!offs || !ptrs) { free(offs); free(ptrs); return PyErr_NoMemory(); }
    int64_t total = 0;
    offs[0] = 0;
    for (Py_ssize_t i = 0; i < n; i++) {
        Py_ssize_t len = 0;
        const char* s = PyUnicode_AsUTF8AndSize(PyList_GET_ITEM(texts, i), &len);  /* not a str: TypeError; a lone surrogate: UnicodeEncodeError */
        if (!s) { free(offs); free(ptrs); return NULL; }
        const char* z = memchr(s, 0, (size_t)len);
        if (z) len = (Py_ssize_t)(z - s);
        ptrs[i] = s;
        total += len;
        offs[i + 1] = total;
    }
    const int64_t cap = hutk_ids_capacity(ctx, total, n);
    uint8_t* bytes = malloc((size_t)(total + 64));
    int32_t* ids = malloc(sizeof(int32_t) * (size_t)(cap > 0 ? cap : 1));
    int64_t* oo = malloc(sizeof(int64_t) * (size_t)(n + 1));
    if (!bytes || !ids || !oo) {
        free(offs); free(ptrs); free(bytes); free(ids); free(oo);
        return PyErr_NoMemory();
    }
    for (Py_ssize_t i = 0; i < n; i++) memcpy(bytes + offs[i], ptrs[i], (size_t)(offs[i + 1] - offs[i]));
    int rc;
    Py_BEGIN_ALLOW_THREADS
    rc = hutk_encode_batch(ctx, bytes, offs, n, ids, cap, oo, NULL);
    Py_END_ALLOW_THREADS
    PyObject* out = NULL;
    if (rc != HUTK_OK && rc != HUTK_E_WORD_TOO_LARGE) {  /* an over-long word ends its document silently (core.c:503) */
        raise_code(rc);
    } else {
        out = PyList_New(n);
        for (Py_ssize_t i = 0; out && i < n; i++) {
            PyObject* e = ids_to_list(ids + oo[i], oo[i + 1] - oo[i]);
            if (!e) { Py_CLEAR(out); break; }
            PyList_SET_ITEM(out, i, e);
        }
    }
    free(offs); free(ptrs); free(bytes); free(ids); free(oo);
    return out;
}

/* list[int] -> int32 array (the reference's (int)PyLong_AsLong, lib.c:915) */
static int32_t* tokens_of(PyObject* list, int64_t* n_out) {
    const Py_ssize_t n = PyList_GET_SIZE(list);
    int32_t* a = malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
    if (!a) { PyErr_NoMemory(); return NULL; }
    for (Py_ssize_t i = 0; i < n; i++) {
        const long v = PyLong_AsLong(PyList_GET_ITEM(list, i));
        if (v == -1 && PyErr_Occurred()) { free(a); return NULL; }
        a[i] = (int32_t)v;
    }
    *n_out = n;
    return a;
}

static PyObject* text_of(const uint8_t* p, int64_t n) {  /* PyUnicode_FromString: ends at the first NUL, strict UTF-8 */
    const uint8_t* z = memchr(p, 0, (size_t)n);
    return PyUnicode_DecodeUTF8((const char*)p, z ? (Py_ssize_t)(z - p) : (Py_ssize_t)n, NULL);
}

static PyObject* decode_docs(hutk_ctx* ctx, const int32_t* ids, const int64_t* id_offs, int64_t n_docs) {
    int64_t* oo = malloc(sizeof(int64_t) * (size_t)(n_docs + 1));
    if (!oo) return PyErr_NoMemory();
    int rc;
    Py_BEGIN_ALLOW_THREADS
    rc = hutk_decode_batch(ctx, ids, id_offs, n_docs, NULL, 0, oo, NULL);  /* sizes */
    Py_END_ALLOW_THREADS
    if (rc != HUTK_OK) { free(oo); return raise_code(rc); }
    const int64_t total = oo[n_docs];
    uint8_t* bytes = malloc((size_t)(total + 16));
    if (!bytes) { free(oo); return PyErr_NoMemory(); }
    Py_BEGIN_ALLOW_THREADS
    rc = hutk_decode_batch(ctx, ids, id_offs, n_docs, bytes, total, oo, NULL);
    Py_END_ALLOW_THREADS
    PyObject* out = NULL;
    if (rc != HUTK_OK) {
        raise_code(rc);
    } else {
        out = PyList_New((Py_ssize_t)n_docs);
        for (int64_t d = 0; out && d < n_docs; d++) {
            PyObject* s = text_of(bytes + oo[d], oo[d + 1] - oo[d]);
            if (!s) { Py_CLEAR(out); break; }
            PyList_SET_ITEM(out, (Py_ssize_t)d, s);
        }
    }
    free(bytes);
    free(oo);
    return out;
}

static PyObject* decode_on(hutk_ctx* ctx, PyObject* args) {
    PyObject* tokens = NULL;
    if (!PyArg_ParseTuple(args, "O", &tokens)) {
        PyErr_SetString(PyExc_TypeError, "Failed to parse arguments. Expected a single list of tokens.");
        return NULL;
    }
    if (!PyList_Check(tokens)) {
        PyErr_SetString(PyExc_TypeError, "Argument must be a list of integers");
        return NULL;
    }
    int64_t n = 0;
    int32_t* ids = tokens_of(tokens, &n);
    if (!ids) return NULL;
    const int64_t offs[2] = {0, n};
    PyObject* l = decode_docs(ctx, ids, offs, 1);
    free(ids);
    if (!l) return NULL;
    PyObject* s = PyList_GET_ITEM(l, 0);
    Py_INCREF(s);
    Py_DECREF(l);
    return s;
}

static PyObject* batch_decode_on(hutk_ctx* ctx, PyObject* args) {
    PyObject* tokens = NULL;
    int num_threads = 1;
    if (!PyArg_ParseTuple(args, "O|i", &tokens, &num_threads) || !PyList_Check(tokens)) {
        PyErr_SetString(PyExc_TypeError, "Failed to parse arguments. Expected a single list of tokens.");
        return NULL;
    }
    const Py_ssize_t n = PyList_GET_SIZE(tokens);
    if (n <= 0) {
        PyErr_SetString(PyExc_ValueError, "No tokens provided.");
        return NULL;
    }
    int64_t* offs = malloc(sizeof(int64_t) * (size_t)(n + 1));
    if (!offs) return PyErr_NoMemory();
    int64_t total = 0;
    offs[0] = 0;
    for (Py_ssize_t i = 0; i < n; i++) {
        PyObject* item = PyList_GET_ITEM(tokens, i);
        if (!PyList_Check(item)) {
            free(offs);
            PyErr_SetString(PyExc_TypeError, "Each item must be a list of integers.");
            return NULL;
        }
        total += PyList_GET_SIZE(item);
        offs[i + 1] = total;
    }
    int32_t* ids = malloc(sizeof(int32_t) * (size_t)(total ? total : 1));
    if (!ids) { free(offs); return PyErr_NoMemory(); }
    for (Py_ssize_t i = 0; i < n; i++) {
        PyObject* item = PyList_GET_ITEM(tokens, i);
        const Py_ssize_t k = PyList_GET_SIZE(item);
        for (Py_ssize_t j = 0; j < k; j++) {
            const long v = PyLong_AsLong(PyList_GET_ITEM(item, j));
            if (v == -1 && PyErr_Occurred()) { free(offs); free(ids); return NULL; }
            ids[offs[i] + j] = (int32_t)v;
        }
    }
    PyObject* out = decode_docs(ctx, ids, offs, n);
    free(offs);
    free(ids);
    return out;
}

/* the method table's entries: take the current context, run on it, let go of it */
#define HUTK_ON_CONTEXT(NAME, IMPL, NOT_INIT)                       \
    static PyObject* NAME(PyObject* self, PyObject* args) {         \
        (void)self;                                                 \
        ctx_box* box = box_acquire();                               \
        if (!box) {                                                 \
            PyErr_SetString(PyExc_RuntimeError, NOT_INIT);          \
            return NULL;                                            \
        }                                                           \
        PyObject* out = IMPL(box->ctx, args);                       \
        box_release(box);                                           \
        return out;                                                 \
    }
HUTK_ON_CONTEXT(p_encode, encode_on, NOT_INIT_ENCODE)
HUTK_ON_CONTEXT(p_batch_encode, batch_encode_on, NOT_INIT_ENCODE)
HUTK_ON_CONTEXT(p_decode, decode_on, NOT_INIT_DECODE)
HUTK_ON_CONTEXT(p_batch_decode, batch_decode_on, NOT_INIT_DECODE)

static PyMethodDef Methods[] = {
    {"initialize", (PyCFunction)p_initialize, METH_VARARGS | METH_KEYWORDS, "Initialize tokenizer"},
    {"encode", (PyCFunction)p_encode, METH_VARARGS, "Encodes string"},
    {"batch_encode", (PyCFunction)p_batch_encode, METH_VARARGS, "Encodes list of strings"},
    {"decode", p_decode, METH_VARARGS, "Decodes list of ints"},
    {"batch_decode", p_batch_decode, METH_VARARGS, "Decodes list of lists of ints"},
    {"handle", p_handle, METH_NOARGS, "Address of the module-global hutk_ctx (0 before initialize)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef Module = {PyModuleDef_HEAD_INIT, "_hutoken_amd",
                                    "huToken's _hutoken method table on the MI355X-native encode path", -1, Methods,
                                    NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__hutoken_amd(void) { return PyModule_Create(&Module); }
