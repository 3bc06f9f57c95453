// train_vocab.cpp -- byte-level BPE trainer used ONCE to produce the frozen
// GPT-2-shaped synthetic vocabulary data/vg50257_*.txt.gz (test/bench data
// infrastructure, not part of the product path).
//
//   train_vocab <kind 2|3|5> <seed> <n_docs> <n_merges> <out.txt> [bytes|chars] [pairs_out.txt]
//   train_vocab 0 <text file> 0 <n_merges> <out.txt> [bytes|chars] [pairs_out.txt]     one document per line of the file
//
// "bytes" (default): initial symbols are the 256 byte values (GPT-2 shape).
// "chars": ' ' is rewritten to U+2581 and initial symbols are whole UTF-8
// characters (SentencePiece/Llama shape); the base characters are written
// first, one per line, before the merges; words holding control bytes are
// skipped (they are one-unit words in the reference's splitter).
//
// Words come from the frozen corpus generator (hutoken_amd/csrc/hutk_synth.c)
// split with the oracle's splitter; output is one line of hex bytes per merge,
// in merge order.  Deterministic: ties go to the smaller (left, right) pair.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <queue>
#include <string>
#include <unordered_map>
#include <vector>

extern "C" {
int64_t hutk_synth_corpus(int kind, uint64_t seed, int64_t first_doc, int64_t n_docs,
                          int num_threads, uint8_t** bytes_out, int64_t* offsets);
void hutk_synth_free(void* p);
size_t hto_split_words(const uint8_t* text, size_t len, uint32_t* starts, size_t cap);
}

struct Word {
    std::vector<int32_t> sym;
    int64_t count;
};

static inline uint64_t key(int32_t a, int32_t b) { return ((uint64_t)(uint32_t)a << 32) | (uint32_t)b; }

struct HeapItem {
    int64_t count;
    uint64_t pair;
    bool operator<(const HeapItem& o) const {
        if (count != o.count) return count < o.count;
        return pair > o.pair;  // smaller pair first
    }
};

int main(int argc, char** argv) {
    if (argc < 6) {
        fprintf(stderr, "usage: %s kind seed n_docs n_merges out\n", argv[0]);
        return 2;
    }
    int kind = atoi(argv[1]);
    uint64_t seed = strtoull(argv[2], nullptr, 0);
    int64_t n_docs = atoll(argv[3]);
    int n_merges = atoi(argv[4]);
    bool chars = argc > 6 && !strcmp(argv[6], "chars");
    // optional 8th argument: file that receives one "LEFTHEX RIGHTHEX" line per merge
    FILE* pairs_out = argc > 7 ? fopen(argv[7], "w") : nullptr;

    std::vector<int64_t> offs(n_docs + 1);
    uint8_t* bytes = nullptr;
    int64_t total = 0;
    std::vector<uint8_t> file_bytes;
    if (kind == 0) {  // a text file, one document per line (the line feeds are not part of the documents)
        FILE* f = fopen(argv[2], "rb");
        if (!f) return 1;
        uint8_t buf[1 << 16];
        size_t got;
        while ((got = fread(buf, 1, sizeof buf, f)) > 0) file_bytes.insert(file_bytes.end(), buf, buf + got);
        fclose(f);
        offs.assign(1, 0);
        std::vector<uint8_t> packed;
        for (size_t a = 0; a < file_bytes.size();) {
            size_t b = a;
            while (b < file_bytes.size() && file_bytes[b] != '\n') b++;
            packed.insert(packed.end(), file_bytes.begin() + a, file_bytes.begin() + b);
            offs.push_back((int64_t)packed.size());
            a = b + 1;
        }
        file_bytes.swap(packed);
        n_docs = (int64_t)offs.size() - 1;
        bytes = file_bytes.data();
        total = (int64_t)file_bytes.size();
    } else {
        total = hutk_synth_corpus(kind, seed, 0, n_docs, 8, &bytes, offs.data());
    }
    if (total < 0) return 1;
    fprintf(stderr, "sample: %lld docs, %lld bytes\n", (long long)n_docs, (long long)total);

    std::unordered_map<std::string, int64_t> wc;
    std::vector<uint32_t> starts;
    for (int64_t d = 0; d < n_docs; d++) {
        const uint8_t* t = bytes + offs[d];
        size_t len = (size_t)(offs[d + 1] - offs[d]);
        starts.resize(len + 1);
        size_t nw = hto_split_words(t, len, starts.data(), starts.size());
        for (size_t w = 0; w < nw; w++) {
            size_t a = starts[w], b = (w + 1 < nw) ? starts[w + 1] : len;
            wc[std::string((const char*)t + a, b - a)]++;
        }
    }
    if (kind != 0) hutk_synth_free(bytes);
    std::vector<Word> words;
    words.reserve(wc.size());
    {
        std::vector<std::pair<std::string, int64_t>> sorted(wc.begin(), wc.end());
        std::sort(sorted.begin(), sorted.end());
        for (auto& kv : sorted) {
            Word w;
            w.count = kv.second;
            for (unsigned char c : kv.first) w.sym.push_back(c);
            words.push_back(std::move(w));
        }
    }
    fprintf(stderr, "unique words: %zu\n", words.size());

    std::vector<std::string> tok(256);
    for (int i = 0; i < 256; i++) tok[i] = std::string(1, (char)i);
    FILE* out = fopen(argv[5], "w");
    if (!out) return 1;
    if (chars) {
        // re-symbolise every word as UTF-8 characters
        std::unordered_map<std::string, int32_t> cid;
        std::vector<std::pair<std::string, int64_t>> ws;
        for (auto& w : words) {
            std::string s;
            bool skip = false;
            for (int32_t c : w.sym) {
                if (c < 32 || c == 127) skip = true;
                if (c == ' ') s += "\xE2\x96\x81"; else s.push_back((char)c);
            }
            if (!skip) ws.push_back({s, w.count});
        }
        std::vector<std::string> base;
        for (auto& kv : ws) {
            const std::string& s = kv.first;
            for (size_t i = 0; i < s.size();) {
                unsigned char b = (unsigned char)s[i];
                size_t l = b < 0x80 ? 1 : (b & 0xE0) == 0xC0 ? 2 : (b & 0xF0) == 0xE0 ? 3 : 4;
                std::string ch = s.substr(i, l);
                if (!cid.count(ch)) { cid[ch] = 0; base.push_back(ch); }
                i += l;
            }
        }
        std::sort(base.begin(), base.end());
        tok.clear();
        for (auto& ch : base) { cid[ch] = (int32_t)tok.size(); tok.push_back(ch); }
        for (auto& ch : base) {
            for (unsigned char c : ch) fprintf(out, "%02X", c);
            fputc('\n', out);
        }
        fprintf(out, "--\n");
        words.clear();
        for (auto& kv : ws) {
            Word w;
            w.count = kv.second;
            const std::string& s = kv.first;
            for (size_t i = 0; i < s.size();) {
                unsigned char b = (unsigned char)s[i];
                size_t l = b < 0x80 ? 1 : (b & 0xE0) == 0xC0 ? 2 : (b & 0xF0) == 0xE0 ? 3 : 4;
                w.sym.push_back(cid[s.substr(i, l)]);
                i += l;
            }
            words.push_back(std::move(w));
        }
        fprintf(stderr, "chars mode: %zu base characters\n", base.size());
    }

    std::unordered_map<uint64_t, int64_t> pc;
    std::unordered_map<uint64_t, std::vector<int32_t>> where;
    for (size_t wi = 0; wi < words.size(); wi++) {
        auto& s = words[wi].sym;
        for (size_t i = 0; i + 1 < s.size(); i++) {
            uint64_t k = key(s[i], s[i + 1]);
            pc[k] += words[wi].count;
            auto& v = where[k];
            if (v.empty() || v.back() != (int32_t)wi) v.push_back((int32_t)wi);
        }
    }
    std::priority_queue<HeapItem> heap;
    for (auto& kv : pc) heap.push({kv.second, kv.first});

    int done = 0;
    while (done < n_merges && !heap.empty()) {
        HeapItem top = heap.top();
        heap.pop();
        auto it = pc.find(top.pair);
        if (it == pc.end() || it->second != top.count) continue;  // stale
        if (top.count < 1) break;
        int32_t a = (int32_t)(top.pair >> 32), b = (int32_t)(top.pair & 0xFFFFFFFFu);
        int32_t nid = (int32_t)tok.size();
        tok.push_back(tok[a] + tok[b]);
        for (unsigned char c : tok.back()) fprintf(out, "%02X", c);
        fputc('\n', out);
        if (pairs_out) {  // the rule itself, for a merges file: left and right halves in hex
            for (unsigned char c : tok[a]) fprintf(pairs_out, "%02X", c);
            fputc(' ', pairs_out);
            for (unsigned char c : tok[b]) fprintf(pairs_out, "%02X", c);
            fputc('\n', pairs_out);
        }
        done++;
        std::vector<int32_t> occ;
        occ.swap(where[top.pair]);
        where.erase(top.pair);
        pc.erase(top.pair);
        std::unordered_map<uint64_t, int64_t> touched;
        for (int32_t wi : occ) {
            auto& s = words[wi].sym;
            int64_t c = words[wi].count;
            std::vector<int32_t> ns;
            ns.reserve(s.size());
            bool changed = false;
            for (size_t i = 0; i < s.size();) {
                if (i + 1 < s.size() && s[i] == a && s[i + 1] == b) {
                    ns.push_back(nid);
                    i += 2;
                    changed = true;
                } else {
                    ns.push_back(s[i]);
                    i++;
                }
            }
            if (!changed) continue;
            for (size_t i = 0; i + 1 < s.size(); i++) touched[key(s[i], s[i + 1])] -= c;
            for (size_t i = 0; i + 1 < ns.size(); i++) {
                uint64_t k = key(ns[i], ns[i + 1]);
                touched[k] += c;
                if (ns[i] == nid || ns[i + 1] == nid) {
                    auto& v = where[k];
                    if (v.empty() || v.back() != wi) v.push_back(wi);
                }
            }
            s.swap(ns);
        }
        for (auto& kv : touched) {
            if (kv.first == top.pair || kv.second == 0) continue;
            int64_t& v = pc[kv.first];
            v += kv.second;
            if (v <= 0) pc.erase(kv.first);
            else heap.push({v, kv.first});
        }
        if (done % 5000 == 0) fprintf(stderr, "merges: %d (last count %lld)\n", done, (long long)top.count);
    }
    fclose(out);
    if (pairs_out) fclose(pairs_out);
    fprintf(stderr, "done: %d merges\n", done);
    return done == n_merges ? 0 : 3;
}
