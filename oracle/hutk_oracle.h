/* hutk_oracle.h -- CPU restatement of huToken's encode path (TEST INFRASTRUCTURE
 * ONLY; see the header of hutk_oracle.c). */
#ifndef HUTK_ORACLE_H
#define HUTK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hto_ctx hto_ctx;

/* loader error kinds, named after the Python exception the reference raises */
enum {
    HTO_OK = 0,
    HTO_E_FILE_NOT_FOUND = 1, /* FileNotFoundError */
    HTO_E_VALUE = 2,          /* ValueError */
    HTO_E_MEMORY = 3          /* MemoryError */
};

/* per-document status */
enum {
    HTO_DOC_OK = 0,
    HTO_DOC_WORD_TOO_LARGE = 1, /* core.c:402-407 */
    HTO_DOC_INVALID_UTF8 = 2,   /* reference behaviour undefined */
    HTO_DOC_NOMEM = 3,
    HTO_DOC_BAD_PATTERN = 4
};

hto_ctx* hto_create(const char* vocab_path, const char* special_path,
                    const char* prefix, int is_byte_encoder, int* err_kind,
                    char* err, size_t errcap);
void hto_destroy(hto_ctx* c);

/* merges file (lib.c:573-663): switches the context to the id-keyed merge path
 * (core.c:211-337, 457-477) when the file has at least one countable line */
int hto_load_merges(hto_ctx* c, const char* path, char* err, size_t errcap);
int hto_has_merges(const hto_ctx* c);
uint64_t hto_rule_count(const hto_ctx* c);

/* regex pre-token path (core.c:350-360): a POSIX ERE, or NULL for the hand-written splitter */
int hto_set_pattern(hto_ctx* c, const char* pattern);

uint64_t hto_vocab_count(const hto_ctx* c);
int hto_vocab_lookup(const hto_ctx* c, const uint8_t* key, size_t len,
                     int32_t* id);
const char* hto_special(const hto_ctx* c, int idx);

/* word splitter alone: writes up to cap word start offsets, returns the word
 * count */
size_t hto_split_words(const uint8_t* text, size_t len, uint32_t* starts,
                       size_t cap);

/* decode direction: one document's ids -> bytes (malloc'd, release with hto_free) */
enum {
    HTO_DEC_OK = 0,
    HTO_DEC_RANGE = 1,     /* id < 0 or >= number of vocabulary lines: ValueError in the reference */
    HTO_DEC_HOLE = 2,      /* no key has this id: undefined in the reference */
    HTO_DEC_AMBIGUOUS = 3, /* two keys have this id: hash-map order decides in the reference */
    HTO_DEC_NOMEM = 4
};
int hto_decode(const hto_ctx* c, const int32_t* ids, size_t n, uint8_t** out, size_t* out_len);

/* one document; *ids_out is malloc'd (release with hto_free) */
int hto_encode(const hto_ctx* c, const uint8_t* text, size_t len,
               int32_t** ids_out, size_t* n_out);
void hto_free(void* p);

/* packed batch: bytes + offsets[n_docs+1] -> ids (malloc'd) +
 * out_offsets[n_docs+1] + status[n_docs] */
int hto_encode_batch(const hto_ctx* c, const uint8_t* bytes,
                     const int64_t* offsets, int64_t n_docs, int num_threads,
                     int32_t** ids_out, int64_t* out_offsets, int32_t* status);

#ifdef __cplusplus
}
#endif
#endif
