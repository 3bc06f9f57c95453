/*
 * hutk_synth.c -- frozen synthetic corpora for tests and bench.py.
 *
 * Integer arithmetic only (splitmix64 + fixed-point tables), so the bytes are
 * identical on every host.  Every document has its own generator seeded from
 * (seed, document index): any document range can be produced independently,
 * which is how bench.py shards a corpus over ranks without generating all of it.
 *
 * Corpora (SURVEY.md section 8 d):
 *   kind 2 "C2"  ASCII: words drawn Zipf(1.1) from a 32768-entry lexicon,
 *                length ~ lognormal(ln 220, 0.55) clamped to [16, 2048] (mean ~256)
 *   kind 3 "C3"  mixed UTF-8: 70 % ASCII words, 20 % Hungarian words (2-byte
 *                letters), 7 % CJK runs of 1-8 characters (3-byte, one word each),
 *                2 % emoji (4-byte), 1 % NBSP / tab / CRLF; length ~
 *                lognormal(ln 440, 0.55) clamped to [16, 8192] (mean ~512)
 *   kind 5 "C5"  Hungarian-like syllable text, lengths as C3
 * Separators: one space 88 %, ", " or ". " 8 %, "\n" 3 %, two spaces 1 %;
 * 5 % of the ASCII tokens are digit groups.  No 0x00, valid UTF-8 only.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LEX_N 32768
#define HUN_N 8192
#define WORD_MAX 40

static inline uint64_t sm64(uint64_t* s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline uint32_t below(uint64_t* s, uint32_t n) {
    return (uint32_t)(((sm64(s) >> 32) * (uint64_t)n) >> 32);
}

struct lexicon {
    uint8_t len[LEX_N];
    char w[LEX_N][WORD_MAX];
    uint64_t cdf[LEX_N]; /* cumulative Zipf weights */
    int n;
};
static struct lexicon g_lex, g_hun;
static uint32_t g_cjk[512];
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

/* floor(2^10 * k^(1/10)) by bisection on r^10 <= k * 2^100 (fits 128 bits) */
static uint32_t root10_q10(uint32_t k) {
    unsigned __int128 target = (unsigned __int128)k << 100;
    uint32_t lo = 1024, hi = 4096;
    while (lo + 1 < hi) {
        uint32_t mid = (lo + hi) / 2;
        unsigned __int128 p = 1;
        for (int i = 0; i < 10; i++) p *= mid;
        if (p <= target) lo = mid; else hi = mid;
    }
    return lo;
}
/* weight(k) ~ 1 / k^1.1, k = 1-based rank */
static void zipf_cdf(uint64_t* cdf, int n) {
    uint64_t acc = 0;
    for (int k = 1; k <= n; k++) {
        uint64_t w = ((uint64_t)1 << 50) / ((uint64_t)k * root10_q10((uint32_t)k));
        acc += w ? w : 1;
        cdf[k - 1] = acc;
    }
}
static int zipf_pick(const uint64_t* cdf, int n, uint64_t* s) {
    uint64_t r = sm64(s) % cdf[n - 1];
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (cdf[mid] > r) hi = mid; else lo = mid + 1;
    }
    return lo;
}

/* English-like letter frequencies, per mille */
static const char LET[] = "etaoinshrdlcumwfgypbvkjxqz";
static const uint16_t LETW[26] = {127, 91, 82, 75, 70, 67, 63, 61, 60, 43, 40, 28, 28,
                                  24, 24, 22, 20, 20, 19, 15, 10, 8, 2, 2, 1, 1};
static char pick_letter(uint64_t* s) {
    uint32_t r = below(s, 1003), acc = 0;
    for (int i = 0; i < 26; i++) {
        acc += LETW[i];
        if (r < acc) return LET[i];
    }
    return 'e';
}

static const char* HU_V[] = {"a", "e", "i", "o", "u", "\xC3\xA1", "\xC3\xA9", "\xC3\xAD",
                             "\xC3\xB3", "\xC3\xB6", "\xC5\x91", "\xC3\xBA", "\xC3\xBC", "\xC5\xB1"};
static const char* HU_C[] = {"b", "c", "d", "f", "g", "h", "j", "k", "l", "m", "n", "p", "r",
                             "s", "t", "v", "z", "sz", "gy", "ny", "cs", "zs", "ty", "ly"};

static void init_tables(void) {
    uint64_t s = 0x4C455849ull; /* "LEXI" */
    g_lex.n = LEX_N;
    for (int k = 0; k < LEX_N; k++) {
        int lo, hi;
        if (k < 64) { lo = 1; hi = 4; }
        else if (k < 1024) { lo = 2; hi = 7; }
        else { lo = 3; hi = 12; }
        int len = lo + (int)below(&s, (uint32_t)(hi - lo + 1));
        for (int i = 0; i < len; i++) g_lex.w[k][i] = pick_letter(&s);
        if (below(&s, 10) == 0) g_lex.w[k][0] = (char)(g_lex.w[k][0] - 32);
        g_lex.len[k] = (uint8_t)len;
    }
    zipf_cdf(g_lex.cdf, LEX_N);
    s = 0x48554E47ull; /* "HUNG" */
    g_hun.n = HUN_N;
    for (int k = 0; k < HUN_N; k++) {
        int syl = 1 + (int)below(&s, k < 128 ? 2 : 4);
        int len = 0;
        for (int i = 0; i < syl; i++) {
            if (i > 0 || below(&s, 4)) {
                const char* c = HU_C[below(&s, 24)];
                size_t l = strlen(c);
                memcpy(g_hun.w[k] + len, c, l);
                len += (int)l;
            }
            const char* v = HU_V[below(&s, 14)];
            size_t l = strlen(v);
            memcpy(g_hun.w[k] + len, v, l);
            len += (int)l;
            if (below(&s, 3) == 0) {
                const char* c = HU_C[below(&s, 17)];
                l = strlen(c);
                memcpy(g_hun.w[k] + len, c, l);
                len += (int)l;
            }
        }
        g_hun.len[k] = (uint8_t)len;
    }
    zipf_cdf(g_hun.cdf, HUN_N);
    for (int i = 0; i < 512; i++) g_cjk[i] = 0x4E00u + (uint32_t)((i * 37) % 20000);
}

/* exp2 in Q16 of a Q16 argument, integer only (33-entry table of 2^(i/32)) */
static const uint32_t EXP2_TAB[33] = {
    65536, 66971, 68438, 69936, 71468, 73032, 74632, 76266, 77936, 79642, 81386,
    83168, 84990, 86851, 88752, 90696, 92682, 94711, 96785, 98905, 101070, 103283,
    105545, 107856, 110218, 112631, 115098, 117618, 120194, 122825, 125515, 128263, 131072};
static uint64_t exp2_q16(int64_t t) {
    int64_t ip = t >> 16;
    uint32_t fp = (uint32_t)(t & 0xFFFF);
    uint32_t i = fp >> 11, r = fp & 0x7FF;
    uint64_t v = EXP2_TAB[i] + (((uint64_t)(EXP2_TAB[i + 1] - EXP2_TAB[i]) * r) >> 11);
    if (ip >= 0) return v << ip;
    return v >> (-ip);
}
/* median * exp(0.55 * N(0,1)); the normal is a sum of 12 uniforms */
static uint32_t doc_target_len(uint64_t* s, uint32_t median, uint32_t lo, uint32_t hi) {
    int64_t z = -6 * 65536;
    for (int i = 0; i < 12; i++) z += (int64_t)(sm64(s) & 0xFFFF);
    /* sigma / ln 2 = 0.55 / 0.693147 = 0.793484 -> Q16 52002 */
    int64_t t = (z * 52002) >> 16;
    uint64_t len = ((uint64_t)median * exp2_q16(t)) >> 16;
    if (len < lo) len = lo;
    if (len > hi) len = hi;
    return (uint32_t)len;
}

static int put_utf8(char* o, uint32_t cp) {
    if (cp < 0x80) { o[0] = (char)cp; return 1; }
    if (cp < 0x800) { o[0] = (char)(0xC0 | (cp >> 6)); o[1] = (char)(0x80 | (cp & 0x3F)); return 2; }
    if (cp < 0x10000) {
        o[0] = (char)(0xE0 | (cp >> 12)); o[1] = (char)(0x80 | ((cp >> 6) & 0x3F));
        o[2] = (char)(0x80 | (cp & 0x3F)); return 3;
    }
    o[0] = (char)(0xF0 | (cp >> 18)); o[1] = (char)(0x80 | ((cp >> 12) & 0x3F));
    o[2] = (char)(0x80 | ((cp >> 6) & 0x3F)); o[3] = (char)(0x80 | (cp & 0x3F));
    return 4;
}

/* one token (word + separator) into tok, returns its length (<= 96) */
static int gen_token(int kind, uint64_t* s, char* tok) {
    int n = 0;
    uint32_t r = below(s, 100);
    int klass; /* 0 ascii, 1 hungarian, 2 cjk, 3 emoji, 4 odd whitespace */
    if (kind == 2) klass = 0;
    else if (kind == 5) klass = 1;
    else klass = r < 70 ? 0 : r < 90 ? 1 : r < 97 ? 2 : r < 99 ? 3 : 4;
    if (klass == 0) {
        if (below(s, 20) == 0) {
            int d = 1 + (int)below(s, 5);
            for (int i = 0; i < d; i++) tok[n++] = (char)('0' + below(s, 10));
        } else {
            int k = zipf_pick(g_lex.cdf, LEX_N, s);
            memcpy(tok, g_lex.w[k], g_lex.len[k]);
            n = g_lex.len[k];
        }
    } else if (klass == 1) {
        int k = zipf_pick(g_hun.cdf, HUN_N, s);
        memcpy(tok, g_hun.w[k], g_hun.len[k]);
        n = g_hun.len[k];
        if (below(s, 12) == 0 && tok[0] >= 'a' && tok[0] <= 'z') tok[0] = (char)(tok[0] - 32);
    } else if (klass == 2) {
        int c = 1 + (int)below(s, 8);
        for (int i = 0; i < c; i++) n += put_utf8(tok + n, g_cjk[below(s, 512)]);
    } else if (klass == 3) {
        n += put_utf8(tok + n, 0x1F600u + below(s, 80));
    } else {
        uint32_t w = below(s, 3);
        if (w == 0) { tok[n++] = (char)0xC2; tok[n++] = (char)0xA0; }
        else if (w == 1) tok[n++] = '\t';
        else { tok[n++] = '\r'; tok[n++] = '\n'; }
        return n;
    }
    uint32_t q = below(s, 100);
    if (q < 88) tok[n++] = ' ';
    else if (q < 92) { tok[n++] = ','; tok[n++] = ' '; }
    else if (q < 96) { tok[n++] = '.'; tok[n++] = ' '; }
    else if (q < 99) tok[n++] = '\n';
    else { tok[n++] = ' '; tok[n++] = ' '; }
    return n;
}

/* one document into out (capacity >= 8192 + 96); returns its length */
static uint32_t gen_doc(int kind, uint64_t seed, int64_t doc, uint8_t* out) {
    uint64_t s = seed ^ ((uint64_t)doc * 0xD1342543DE82EF95ull);
    sm64(&s);
    uint32_t target = (kind == 2) ? doc_target_len(&s, 220, 16, 2048)
                                  : doc_target_len(&s, 440, 16, 8192);
    uint32_t n = 0;
    char tok[128];
    for (;;) {
        int tl = gen_token(kind, &s, tok);
        if (n + (uint32_t)tl > target) {
            if (n == 0) { memcpy(out, tok, (size_t)tl); n = (uint32_t)tl; }
            break;
        }
        memcpy(out + n, tok, (size_t)tl);
        n += (uint32_t)tl;
    }
    return n;
}

struct job {
    int kind;
    uint64_t seed;
    int64_t first, count;
    uint8_t* buf;
    size_t len, cap;
    int64_t* lens; /* per doc */
};

static void* job_run(void* a) {
    struct job* j = a;
    uint8_t tmp[8192 + 256];
    for (int64_t i = 0; i < j->count; i++) {
        uint32_t n = gen_doc(j->kind, j->seed, j->first + i, tmp);
        if (j->len + n > j->cap) {
            size_t nc = j->cap ? j->cap * 2 : (1u << 20);
            while (nc < j->len + n) nc *= 2;
            uint8_t* nb = realloc(j->buf, nc);
            if (!nb) { j->count = -1; return NULL; }
            j->buf = nb;
            j->cap = nc;
        }
        memcpy(j->buf + j->len, tmp, n);
        j->len += n;
        j->lens[i] = n;
    }
    return NULL;
}

/* Generates documents [first_doc, first_doc + n_docs) of corpus `kind`.
 * offsets must hold n_docs + 1 entries.  *bytes_out is malloc'd; release it
 * with hutk_synth_free.  Returns the total byte count or -1. */
int64_t hutk_synth_corpus(int kind, uint64_t seed, int64_t first_doc, int64_t n_docs,
                          int num_threads, uint8_t** bytes_out, int64_t* offsets) {
    pthread_once(&g_once, init_tables);
    if (kind != 2 && kind != 3 && kind != 5) return -1;
    if (num_threads < 1) num_threads = 1;
    if (num_threads > 64) num_threads = 64;
    if ((int64_t)num_threads > n_docs) num_threads = n_docs > 0 ? (int)n_docs : 1;
    struct job jobs[64];
    pthread_t th[64];
    int64_t* lens = malloc(sizeof(int64_t) * (size_t)(n_docs ? n_docs : 1));
    if (!lens) return -1;
    int64_t per = (n_docs + num_threads - 1) / num_threads;
    for (int t = 0; t < num_threads; t++) {
        int64_t a = per * t, b = a + per;
        if (a > n_docs) a = n_docs;
        if (b > n_docs) b = n_docs;
        jobs[t] = (struct job){kind, seed, first_doc + a, b - a, NULL, 0, 0, lens + a};
        pthread_create(&th[t], NULL, job_run, &jobs[t]);
    }
    int64_t total = 0;
    int bad = 0;
    for (int t = 0; t < num_threads; t++) {
        pthread_join(th[t], NULL);
        if (jobs[t].count < 0) bad = 1;
        total += (int64_t)jobs[t].len;
    }
    uint8_t* all = bad ? NULL : malloc((size_t)(total ? total : 1));
    if (!all) {
        for (int t = 0; t < num_threads; t++) free(jobs[t].buf);
        free(lens);
        return -1;
    }
    size_t at = 0;
    for (int t = 0; t < num_threads; t++) {
        if (jobs[t].len) memcpy(all + at, jobs[t].buf, jobs[t].len);
        at += jobs[t].len;
        free(jobs[t].buf);
    }
    int64_t acc = 0;
    for (int64_t i = 0; i < n_docs; i++) {
        offsets[i] = acc;
        acc += lens[i];
    }
    offsets[n_docs] = acc;
    free(lens);
    *bytes_out = all;
    return total;
}

void hutk_synth_free(void* p) { free(p); }

/* Lengths only (bytes per document) of documents [first_doc, first_doc + n_docs): what a rank needs to find its
 * byte-balanced shard of a corpus without holding the corpus (bench.py, strong scaling). */
struct len_job { int kind; uint64_t seed; int64_t first, count; int64_t* lens; };
static void* len_run(void* a) {
    struct len_job* j = a;
    uint8_t tmp[8192 + 256];
    for (int64_t i = 0; i < j->count; i++) j->lens[i] = gen_doc(j->kind, j->seed, j->first + i, tmp);
    return NULL;
}
int hutk_synth_lengths(int kind, uint64_t seed, int64_t first_doc, int64_t n_docs, int num_threads, int64_t* lens) {
    pthread_once(&g_once, init_tables);
    if (kind != 2 && kind != 3 && kind != 5) return -1;
    if (num_threads < 1) num_threads = 1;
    if (num_threads > 64) num_threads = 64;
    struct len_job jobs[64];
    pthread_t th[64];
    int64_t per = (n_docs + num_threads - 1) / num_threads;
    for (int t = 0; t < num_threads; t++) {
        int64_t a = per * t, b = a + per;
        if (a > n_docs) a = n_docs;
        if (b > n_docs) b = n_docs;
        jobs[t] = (struct len_job){kind, seed, first_doc + a, b - a, lens + a};
        pthread_create(&th[t], NULL, len_run, &jobs[t]);
    }
    for (int t = 0; t < num_threads; t++) pthread_join(th[t], NULL);
    return 0;
}
