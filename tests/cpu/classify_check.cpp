// classify_check.cpp -- host fuzz of hutk_classify.h (the device word splitter)
// against the oracle's sequential splitter (oracle/hutk_oracle.c).  Built and run by
// tests/test_classify_cpu.py.   usage: classify_check <n_cases> <seed> [mode]
//                                        classify_check --texts   (hex text per stdin line -> its word starts by every form)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hutk_classify.h"

extern "C" size_t hto_split_words(const uint8_t* text, size_t len, uint32_t* starts, size_t cap);

static uint64_t rng_s;
static uint64_t rnd() {
    uint64_t z = (rng_s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static const char* PIECES[] = {
    "a", "b", "Z", "e", "t", " ", " ", " ", "  ", "   ", "1", "42", ".", ",", "!", "\t", "\n", "\r\n", "\x0b", "\x0c",
    "\x01", "\x7f", "\xc3\xa9", "\xc3\xa1", "\xc5\x91", "\xc5\xb1", "\xc3\x96", "\xc3\xa4", "\xc5\x82", "\xc2\xa0",
    "\xc2\x85", "\xe6\xbc\xa2", "\xe5\xad\x97", "\xe2\x82\xac", "\xe0\xa4\x95", "\xe0\xb8\x81", "\xed\xa0\x80",
    "\xf0\x9f\x98\x82", "\xf0\x90\x8d\x88", "\xf4\x8f\xbf\xbf", "\xf5\x80\x80\x80", "\xc3", "\xe6\xbc", "\xf0\x9f",
    "\xf0\x9f\x98", "\x80", "\xbf", "\xa0", "\xff", "\xf8", "\xfe", "\xc0\xa0", "\xc0\x80", "\xc1\xa1", "\xc1\x81",
    "\xc0\xb0", "\xc0\x89", "\xe0\x80\x80", "\xe0\x80\xa0", "\xe0\x81\xa1", "\xe0\x83\xa1", "\xe0\x9f\xbf",
    "\xf0\x80\x80\x80", "\xf0\x80\x80\xa0", "\xf0\x80\x81\xa1", "\xf0\x8f\xbf\xbf", "\xc3\xc3", "\xe6\x20",
    "word", "Hello", "\xc3\xa1rv\xc3\xadzt\xc5\xb1r\xc5\x91"};

// word starts of one document by the three forms of the device splitter, 16 positions per call as the kernel does
static void starts_of(const std::vector<uint8_t>& text, const uint16_t* dfa_table, const uint8_t* dfa_lut,
                      std::vector<uint32_t> out[3]) {
    const size_t n = text.size();
    for (size_t p0 = 0; p0 < n; p0 += 16) {
        uint32_t d[8] = {0};
        uint32_t dbits = 0;
        for (int k = 0; k < 32; k++) {
            const long p = (long)p0 - 8 + k;
            if (p >= 0 && p < (long)n) d[k >> 2] |= (uint32_t)text[p] << (8 * (k & 3));
            if (p == 0 || p == (long)n) dbits |= 1u << k;
        }
        bool ex_s = false, ex_d = false;
        const uint32_t fe = hutk::classify16_exact(d, dbits);
        uint32_t fs = hutk::classify16(d, dbits, &ex_s), fd = hutk::classify16_dfa(d, dbits, dfa_table, dfa_lut, &ex_d);
        if (ex_s) fs = fe;  // (what the kernel does with a window the fast forms hand back)
        if (ex_d) fd = fe;
        {
            bool ex_d2 = false;
            uint32_t fd2 = hutk::classify16_dfa2(d, dbits, dfa_table, dfa_lut, &ex_d2);
            if (ex_d2) fd2 = fe;
            if (fd2 != fd) { fprintf(stderr, "two-walk automaton differs\n"); abort(); }
        }
        for (int j = 0; j < 16 && p0 + j < n; j++) {
            if ((fe >> j) & 1u) out[0].push_back((uint32_t)(p0 + j));
            if ((fs >> j) & 1u) out[1].push_back((uint32_t)(p0 + j));
            if ((fd >> j) & 1u) out[2].push_back((uint32_t)(p0 + j));
        }
    }
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "--texts")) {
        static uint16_t tab[hutk::dfa::TABLE_BYTES / 2];
        static uint8_t lut[256];
        hutk::dfa::build(tab, lut);
        char line[1 << 16];
        while (fgets(line, sizeof line, stdin)) {
            std::vector<uint8_t> text;
            for (char* c = line; c[0] && c[1] && c[0] != '\n'; c += 2) {
                unsigned v;
                if (sscanf(c, "%2x", &v) != 1) break;
                text.push_back((uint8_t)v);
            }
            std::vector<uint32_t> out[3];
            starts_of(text, tab, lut, out);
            for (int f = 0; f < 3; f++) {
                printf("%s", f ? ";" : "");
                for (size_t i = 0; i < out[f].size(); i++) printf("%s%u", i ? "," : "", out[f][i]);
            }
            printf("\n");
        }
        return 0;
    }
    const long n_cases = argc > 1 ? atol(argv[1]) : 20000;
    rng_s = argc > 2 ? strtoull(argv[2], nullptr, 0) : 1;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;  // 1: no overlong pieces (keeps the fast paths in play); 2: byte soup
    const int NP = (int)(sizeof(PIECES) / sizeof(PIECES[0]));
    auto overlong = [](const char* pc) {
        const unsigned char* u = (const unsigned char*)pc;
        for (; *u; u++)
            if (*u == 0xC0 || *u == 0xC1 || (*u == 0xE0 && u[1] < 0xA0) || (*u == 0xF0 && u[1] < 0x90)) return true;
        return false;
    };
    long bad = 0, exotic_windows = 0, windows = 0, dfa_exotic = 0;
    static uint16_t dfa_table[hutk::dfa::TABLE_BYTES / 2];
    static uint8_t dfa_lut[256];
    hutk::dfa::build(dfa_table, dfa_lut);
    for (long cs = 0; cs < n_cases; cs++) {
        // a batch of a few documents packed back to back
        std::vector<uint8_t> text;
        std::vector<size_t> offs{0};
        const int n_docs = 1 + (int)(rnd() % 4);
        for (int dd = 0; dd < n_docs; dd++) {
            const int np = (int)(rnd() % (mode == 2 ? 40 : 14));
            const bool ascii_only = rnd() % 4 == 0;
            for (int i = 0; i < np; i++) {
                if (mode == 2) {  // single bytes: leads, continuation bytes and ASCII in any order (no overlong leads)
                    static const uint8_t B[] = {' ', ' ', 'a', 'b', '1', '.', '\n', 0xC2, 0xC3, 0xC3, 0xC5, 0xC5, 0xD0, 0xE1, 0xE2,
                                                0xED, 0xF1, 0xF4, 0xF7, 0xF8, 0x80, 0x81, 0x90, 0x91, 0x96, 0xA0, 0xA1, 0xA9,
                                                0xB0, 0xB1, 0xBF, 0x9C, 0xBC, 0x8D};
                    text.push_back(B[rnd() % sizeof(B)]);
                    continue;
                }
                const char* pc = PIECES[rnd() % (ascii_only ? 22 : NP)];
                if (mode == 1 && overlong(pc)) { i--; continue; }
                text.insert(text.end(), pc, pc + strlen(pc));
            }
            offs.push_back(text.size());
        }
        const size_t n = text.size();
        std::vector<uint8_t> expect(n + 1, 0);
        std::vector<uint32_t> starts(n + 1);
        for (int dd = 0; dd < n_docs; dd++) {
            const size_t a = offs[dd], b = offs[dd + 1];
            const size_t nw = hto_split_words(text.data() + a, b - a, starts.data(), starts.size());
            for (size_t w = 0; w < nw; w++) expect[a + starts[w]] = 1;
        }
        std::vector<uint8_t> docstart(n + 1, 0);
        for (size_t o : offs) docstart[o] = 1;  // includes the end sentinel
        for (size_t p0 = 0; p0 < n; p0 += 16) {
            uint32_t d[8] = {0};
            uint32_t dbits = 0;
            for (int k = 0; k < 32; k++) {
                const long p = (long)p0 - 8 + k;
                if (p >= 0 && p < (long)n) d[k >> 2] |= (uint32_t)text[p] << (8 * (k & 3));
                if (p >= 0 && p <= (long)n && docstart[p]) dbits |= 1u << k;
            }
            bool exotic = false;
            const uint32_t fs = hutk::classify16(d, dbits, &exotic);
            const uint32_t fe = hutk::classify16_exact(d, dbits);
            bool exotic_d = false;
            const uint32_t fd = hutk::classify16_dfa(d, dbits, dfa_table, dfa_lut, &exotic_d);
            dfa_exotic += exotic_d;
            bool exotic_d2 = false;  // the two-walk form (k_ptiles): the same answer wherever it does not hand the window back
            const uint32_t fd2 = hutk::classify16_dfa2(d, dbits, dfa_table, dfa_lut, &exotic_d2);
            if (!exotic_d2) {
                uint32_t m = 0;
                for (int j = 0; j < 16 && p0 + j < n; j++) m |= (uint32_t)expect[p0 + j] << j;
                const uint32_t valid = (n - p0 >= 16) ? 0xFFFFu : ((1u << (n - p0)) - 1u);
                if ((fd2 & valid) != m) {
                    if (bad < 10) fprintf(stderr, "MISMATCH (two-walk automaton) case %ld pos %zu: want %04x got %04x\n", cs, p0, m, fd2 & valid);
                    bad++;
                }
            }
            windows++;
            exotic_windows += exotic;
            for (int j = 0; j < 16 && p0 + j < n; j++) {
                const uint32_t want = expect[p0 + j];
                const uint32_t got_e = (fe >> j) & 1u, got_s = (fs >> j) & 1u, got_d = (fd >> j) & 1u;
                if (got_e != want || (!exotic && got_s != want) || (!exotic_d && got_d != want)) {
                    if (bad < 10) {
                        fprintf(stderr, "MISMATCH case %ld pos %zu: want %u exact %u swar %u exotic %d dfa %u exotic %d  text:",
                                cs, p0 + j, want, got_e, got_s, (int)exotic, got_d, (int)exotic_d);
                        for (size_t q = 0; q < n; q++) fprintf(stderr, " %02x%s", text[q], docstart[q + 1] ? " |" : "");
                        fprintf(stderr, "\n");
                    }
                    bad++;
                }
            }
        }
    }
    printf("cases %ld windows %ld exotic %ld dfa-exotic %ld mismatches %ld\n", n_cases, windows, exotic_windows,
           dfa_exotic, bad);
    return bad ? 1 : 0;
}
