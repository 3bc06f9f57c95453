/*
 * hutoken_amd.h -- C ABI of the MI355X-native batch BPE encode path.
 *
 * This is the drop-in boundary for huToken's encode direction.  Each entry point
 * names the reference interface it replaces (paths are into the reference tree,
 * matyasosvath/hutoken @ 2025-09-05).  Plain pointers and sizes only; no Python,
 * torch or HIP types appear in a signature (a HIP stream is passed as void*).
 * INTEGRATION.md shows the binding a maintainer would add to src/lib.c.
 *
 * All results are bit-exact with the reference's string-keyed path
 * (src/core.c:66-209, 339-511) on the same inputs.  There is no CPU fallback:
 * every encode call runs on the GPU or fails with HUTK_E_DEVICE.
 */
#ifndef HUTOKEN_AMD_H
#define HUTOKEN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hutk_ctx hutk_ctx;

/* return codes; the comment names the Python exception the reference raises in
 * the same situation (src/lib.c) */
enum {
    HUTK_OK = 0,
    HUTK_E_FILE_NOT_FOUND = 1,  /* FileNotFoundError (lib.c:243-250, 460-469) */
    HUTK_E_VALUE = 2,           /* ValueError        (lib.c:295-388, 487-543) */
    HUTK_E_MEMORY = 3,          /* MemoryError */
    HUTK_E_ARG = 4,             /* TypeError / bad argument */
    HUTK_E_DEVICE = 5,          /* no GPU, HIP failure: the path fails loudly */
    HUTK_E_UNSUPPORTED = 6,     /* a file shape the device tables cannot hold */
    HUTK_E_CAPACITY = 7,        /* ids_cap below hutk_ids_capacity() */
    HUTK_E_NUL_BYTE = 8,        /* a 0x00 byte inside a document */
    HUTK_E_WORD_TOO_LARGE = 9,  /* device-side note only, see HUTK_DOC_WORD_TOO_LARGE */
    HUTK_E_INVALID_UTF8 = 10    /* non-byte-encoder mode only; the reference's
                                   behaviour there is undefined */
};

/* per-document status values written to status[] */
enum {
    HUTK_DOC_OK = 0,
    HUTK_DOC_WORD_TOO_LARGE = 1, /* a word longer than 262144 bytes (core.c:402-407).  The
                                    reference reports NOTHING for it (core.c:503 clears the
                                    message again) and ends the document there; so does
                                    hutk_encode_batch: the ids are those before that word */
    HUTK_DOC_INVALID_UTF8 = 2,
    /* decode direction */
    HUTK_DOC_ID_OUT_OF_RANGE = 3, /* an id < 0 or >= the number of vocabulary lines (src/core.c:523-531) */
    HUTK_DOC_ID_UNDECODABLE = 4   /* an id without a unique key, or a token whose decoding depends on its
                                     neighbours (see hutk_decode_batch) */
};

/* Replaces _hutoken.initialize(vocab_file_path, special_file_path, prefix,
 * is_byte_encoder, ...) for the encode direction: src/lib.c:185-571
 * (initialize_context 128-183, vocab loader 243-388, special-character loader
 * 460-571) and struct EncodeContext (include/hutoken/taskqueue.h:16-25).
 * The decode tables are built alongside (hutk_decode_*), the regex pattern is set with
 * hutk_ctx_set_pattern.  `device` is a HIP device ordinal, or -1 for the current
 * device.  On failure *out is NULL and hutk_last_error() holds the message. */
int hutk_ctx_create(hutk_ctx** out, const char* vocab_path, const char* special_path,
                    const char* prefix, int is_byte_encoder, int device);

/* The same with _hutoken.initialize's `merges_file_path` (src/lib.c:573-663): when
 * the file holds at least one countable line the context encodes with the
 * reference's ID-KEYED merge loop (bpe_encode_arena_ids, src/core.c:211-337; unit
 * split and id lookup of src/core.c:457-477): rank = line order among the rules
 * whose left, right and concatenation are vocabulary keys, result = the id of the
 * concatenation, a repeated (left id, right id) keeps its last rule.  A file with
 * no countable line (empty, comments only) leaves the string-keyed path in force,
 * as in the reference.  merges_path == NULL is hutk_ctx_create.  A special-character
 * replacement of more than one character is several units per input item on this path
 * (src/core.c:460-474 splits it per character): accepted, its words take the exception path. */
int hutk_ctx_create_merges(hutk_ctx** out, const char* vocab_path, const char* special_path,
                           const char* prefix, int is_byte_encoder, const char* merges_path,
                           int device);

/* _hutoken.initialize's `pattern` (src/lib.c:188-205, 229-232): the regex pre-token path of encode(), src/core.c:350-360,
 * 372-378, 392-400, 498-500.  `pattern` is a POSIX extended regular expression; the words of a document are the
 * successive LEFTMOST matches at or after a cursor, text between them is dropped, an empty match moves the cursor one
 * byte on.  Matching is libc's regcomp/regexec in the process's locale, exactly the calls the reference makes, run on
 * the host by hutk_encode_batch / hutk_encode (one compiled pattern per host thread); pretokenizer and merge loop stay
 * on the GPU.  NULL returns to the hand-written splitter (src/parser.c).  HUTK_E_VALUE: the pattern does not compile
 * (the reference compares regcomp()'s result with `true` and goes on with an uncompiled pattern for every other error
 * code).  A context with a prefix keeps it: the prefix goes with a document's first match (src/core.c:364-366, 421-451).
 * hutk_encode_batch_device on a context with a pattern copies the bytes down for regexec and SYNCHRONISES with the
 * stream (the only form of the call that does); the encode itself stays on the device buffers. */
int hutk_ctx_set_pattern(hutk_ctx* ctx, const char* pattern);

/* Several GPUs behind ONE context of one process (`device_mask`).  The reference's
 * batch_encode spreads the documents of a batch over host threads, balanced by a DP on their lengths
 * (src/lib.c:779-794, 48-57); here hutk_encode_batch cuts a batch of at least 4 MiB (HUTK_MULTI_MIN_BYTES overrides) into
 * runs of whole documents with about the same number of bytes, one per device, encodes them side by side on a host
 * thread each, and puts the ids together in document order: ids, offsets and status are those of one device.
 * hutk_ctx_add_device adds `device` (an ordinal of this process; the same ordinal may be added again) with the tables
 * the context was created from; everything that takes device pointers (hutk_encode_batch_device,
 * hutk_decode_batch_device) and the decode direction stay on the context's first device. */
int hutk_ctx_add_device(hutk_ctx* ctx, int device);
int hutk_ctx_device_count(const hutk_ctx* ctx); /* 1 + the devices added; 0 for a host-only context */

/* The reference never frees its contexts (lib.c:129-155); this one can be. */
void hutk_ctx_destroy(hutk_ctx* ctx);

/* Message of the last failure on this thread (static storage, never NULL). */
const char* hutk_last_error(void);

/* Worst-case number of ids for a batch of n_bytes bytes in n_docs documents
 * (#ids <= #units <= bytes x the most units one input item can become + prefix units per document).
 * DEVICE MEMORY: beside the caller's buffers a context keeps a workspace that grows to the largest batch it has seen --
 * about 18 bytes per input byte for ordinary vocabulary files.  A special-characters file with a replacement of SEVERAL
 * units (a Llama-style "<0x0A>" on the merges path, test_pretokenizer.c:38-41's 'a' -> "Alpha") makes every word that
 * holds such an item an exception word: the exception arrays are then sized for a word per byte and for that many units
 * per byte, about 60 + 12 x (units per item) bytes of workspace per input byte (a 500 MB batch: tens of GB).  Cut such
 * batches smaller; hutk_encode_batch does so by itself (HUTK_PIPE_CHUNK_MB). */
int64_t hutk_ids_capacity(const hutk_ctx* ctx, int64_t n_bytes, int64_t n_docs);

/* Replaces the worker pool of p_batch_encode, src/lib.c:779-794, i.e. N threads
 * calling `void encode(struct EncodeTask*)` (include/hutoken/core.h:11,
 * src/core.c:339-511) once per document.  Batch-granular: documents are packed
 * back to back in `bytes`; document i is bytes[offsets[i] .. offsets[i+1]).
 * Host buffers in, host buffers out (the copies over PCIe are inside the call).
 *   ids_out      int32[ids_cap], ids_cap >= hutk_ids_capacity(...)
 *   out_offsets  int64[n_docs+1]; document i's ids are
 *                ids_out[out_offsets[i] .. out_offsets[i+1])
 *   status       int32[n_docs] (HUTK_DOC_*), may be NULL
 * Returns HUTK_OK or the first error. */
int hutk_encode_batch(hutk_ctx* ctx, const uint8_t* bytes, const int64_t* offsets,
                      int64_t n_docs, int32_t* ids_out, int64_t ids_cap,
                      int64_t* out_offsets, int32_t* status);

/* THREADS AND STREAMS.  A context owns one workspace (tile metadata, symbol runs, exception lists), so the calls on one
 * context are serialised: on the host by a mutex inside the context (any thread may call), on the device by an event --
 * each asynchronous call records it behind its last kernel and the next call makes its stream wait for it before its
 * first kernel, whichever streams the two calls use.  Nothing races and nothing is freed under a running kernel, but two
 * calls on ONE context do not overlap; give every stream that should encode concurrently a context of its own (the
 * tables are a few MB).  The reference's contexts are read-only and shared by its worker threads (taskqueue.h:16-25); its
 * concurrency is inside one batch_encode call, as is this library's. */

/* Same computation with every buffer already resident in device memory (the
 * form bench.py times and a GPU data loader would call).  n_bytes must equal
 * offsets[n_docs] (the host needs it to size the launch without a sync).
 * Work is enqueued on `hip_stream` (a hipStream_t, NULL = default stream) and
 * the call returns without synchronising (a context with a regex pattern excepted:
 * hutk_ctx_set_pattern); d_err receives the first device-side
 * error code (HUTK_OK when none) and may be NULL.  d_bytes must be 16-byte
 * aligned.  A document with a word of more than 262144 bytes ends in front of that
 * word, as the reference's does (src/core.c:402-407, 503): d_status[i] =
 * HUTK_DOC_WORD_TOO_LARGE, *d_err = HUTK_E_WORD_TOO_LARGE (a note, not a failure),
 * and d_out_offsets / d_ids_out hold the shortened document -- the same as
 * hutk_encode_batch returns. */
int hutk_encode_batch_device(hutk_ctx* ctx, const uint8_t* d_bytes, const int64_t* d_offsets,
                             int64_t n_docs, int64_t n_bytes, int32_t* d_ids_out,
                             int64_t ids_cap, int64_t* d_out_offsets, int32_t* d_status,
                             int32_t* d_err, void* hip_stream);

/* Page-locked host memory for the buffers handed to hutk_encode_batch: with it the chunked
 * path of hutk_encode_batch copies by DMA while the previous chunk is being encoded and the one
 * before is being copied back (pageable buffers work too, at roughly a third of the rate).
 * The pages are taken from the current HIP device's NUMA node where the host has several, provided the calling thread
 * runs under the default memory policy (a policy given with numactl --membind / --interleave is the caller's and stays;
 * HUTK_HOST_ALLOC_NUMA=0: the policy is never touched).  The reference has no counterpart (it strdup()s every text, src/lib.c:770-772). */
void* hutk_host_alloc(size_t n_bytes);
void hutk_host_free(void* p);

/* Replaces p_encode, src/lib.c:668-720 (one document on the calling thread). */
int hutk_encode(hutk_ctx* ctx, const uint8_t* text, int64_t len, int32_t* ids_out,
                int64_t ids_cap, int64_t* n_ids, int32_t* status);

/* Decode direction.  Replaces the worker pool of p_batch_decode / p_decode (src/lib.c:876-1126): one
 * decode(struct DecodeTask*) per document (src/core.c:513-581: token strings concatenated, then
 * pretokenizer_decode, src/pretokenizer.c:197-296: prefix stripped from the front, special values mapped
 * back to their bytes by longest match, other characters to the byte of their code point (byte-encoder
 * mode, '?' above 255) or copied).
 * ids[id_offsets[d] .. id_offsets[d+1]) are the tokens of document d; the decoded BYTES of document d are
 * bytes_out[out_offsets[d] .. out_offsets[d+1]) (the Python layer cuts at the first 0x00 and decodes UTF-8
 * like PyUnicode_FromString, lib.c:938-939).  bytes_out == NULL: only out_offsets (hence the sizes) and
 * status are produced; call again with out_offsets[n_docs] bytes of room.
 * Returns HUTK_E_VALUE when an id is out of range (the reference's ValueError "Element must be
 * non-negative and less than vocab size.") and HUTK_E_UNSUPPORTED when a token cannot be decoded on its
 * own: an id that no key or several keys carry (uninitialised memory / hash-map order in the reference),
 * or a token that ends inside a longer special value or inside a character (its decoding would depend
 * on the next token).  status[] names the documents. */
int hutk_decode_batch(hutk_ctx* ctx, const int32_t* ids, const int64_t* id_offsets, int64_t n_docs,
                      uint8_t* bytes_out, int64_t bytes_cap, int64_t* out_offsets, int32_t* status);
/* The same on device-resident buffers, asynchronously on `hip_stream`; *d_err receives the error code. */
int hutk_decode_batch_device(hutk_ctx* ctx, const int32_t* d_ids, const int64_t* d_id_offsets,
                             int64_t n_docs, int64_t n_ids, uint8_t* d_bytes_out, int64_t bytes_cap,
                             int64_t* d_out_offsets, int32_t* d_status, int32_t* d_err, void* hip_stream);

/* Introspection (tests, bench). */
int64_t hutk_vocab_size(const hutk_ctx* ctx);      /* distinct keys loaded */
int64_t hutk_pair_table_entries(const hutk_ctx* ctx);
int hutk_uses_merges(const hutk_ctx* ctx);         /* 1: the id-keyed merge path is in force */
int hutk_device_ordinal(const hutk_ctx* ctx);
/* out8: distinct keys, vocabulary symbols, symbols, pair entries, pair slots,
 * rank_is_sym, ident_ids, whole-word table entries.  Passing device = -2 to hutk_ctx_create
 * builds a host-only context (tables, no GPU) for this kind of inspection;
 * encode calls on it fail with HUTK_E_DEVICE. */
int hutk_table_stats(const hutk_ctx* ctx, int64_t* out8);

/* Device time of the most recent hutk_encode_batch_device/hutk_encode_batch call,
 * from HIP events recorded on the launch stream: the dominant kernel
 * ("encode tiles") and the whole enqueue.  Synchronises on those events. */
int hutk_last_timing(hutk_ctx* ctx, float* ms_tile_kernel, float* ms_total);

/* Diagnostic: pairs of the (left, right) -> merged table that live in their second bucket (a lookup for them
 * costs two loads instead of one). */
int64_t hutk_debug_pairs_second(const hutk_ctx* ctx);

/* Diagnostic: entries of the whole-word table's companion for words of 15..28 bytes (13..28 with 32-bit symbols). */
int64_t hutk_debug_long_words(const hutk_ctx* ctx);

/* Diagnostic: the seam map the tile kernel splits words by.  Bit (y - 0xE0) of out256[x] is set when some merge of
 * this vocabulary can join a token that ends with input byte x to one that begins with input byte y (0xE0..0xFF, the
 * lead bytes of three- and four-byte characters); where it is clear the reference's merge loop (src/core.c:66-209,
 * 211-337) can never produce a token across x | y, and the word is encoded as two.  Returns 1 when the map is in
 * use, 0 when it is switched off (HUTK_NO_SEAM=1). */
int hutk_debug_seam(const hutk_ctx* ctx, uint32_t* out256);
/* Diagnostic: the seam map's second level (vocabularies whose merges cover every (last byte, lead byte) pair but join only
 * some pairs of whole characters).  a3, b3: the three bytes of the character in front of a boundary and of the one behind
 * it, little-endian.  Returns 1 when no token can span a3 | b3 -- the tile kernel starts a word at b3 although the first
 * level says "may join" -- and 0 otherwise (also when the level is off).  tests/test_seam_cpu.py. */
int hutk_debug_seam2_cut(const hutk_ctx* ctx, uint32_t a3, uint32_t b3);

/* Diagnostic build aid: clock64 stamps at the phase boundaries of the tile kernel.
 * hutk_debug_profile(ctx, 1), run a batch, then hutk_debug_profile_read returns the
 * mean shader cycles per phase over the first n_tiles tiles (out10[0] = their sum,
 * out10[k] = phase k).  Never enabled in timed runs. */
int hutk_debug_profile(hutk_ctx* ctx, int enable);
int hutk_debug_tile_bytes(void); /* input bytes per tile of the hot kernel */
int hutk_debug_profile_read(hutk_ctx* ctx, int64_t n_tiles, double* out10);
int hutk_debug_profile_raw(hutk_ctx* ctx, int64_t n_tiles, long long* out); /* the stamps, ten per tile */

/* Per-call profiling events cost a little; they are on by default. */
void hutk_set_timing(hutk_ctx* ctx, int enabled);

#ifdef __cplusplus
}
#endif
#endif /* HUTOKEN_AMD_H */
