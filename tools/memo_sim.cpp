// memo_sim.cpp -- CPU model of the batch-scoped merge memo of k_tiles ("memo"): what share of
// the merge-loop words of a COLD batch would find their encoding in a memo that the batch itself fills, given that
// ~2048 workgroups run at the same time and see only what earlier ones inserted.  Measurement tool; uses the oracle
// (test infrastructure) to classify words; never part of the product.
//
//   g++ -O2 -std=c++17 tools/memo_sim.cpp -o /tmp/memo_sim -Ioracle -Loracle/_build -lhutk_oracle
//       -Lhutoken_amd/lib -lhutk_synth -Wl,-rpath,$PWD/oracle/_build -Wl,-rpath,$PWD/hutoken_amd/lib
//   /tmp/memo_sim <vocab> <special> <kind 2|3|5> <n_docs> [key_bytes=16] [max_tokens=8] [byte_encoder=1] [prefix]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>
#include <algorithm>

#include "hutk_oracle.h"

extern "C" int64_t hutk_synth_corpus(int kind, uint64_t seed, int64_t first_doc, int64_t n_docs, int num_threads,
                                     uint8_t** bytes_out, int64_t* offsets);

struct Info { uint16_t ntok, nunits; int32_t first_round; int32_t seen_round; uint64_t h; };

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    const int kind = atoi(argv[3]);
    const int64_t n_docs = atoll(argv[4]);
    const int key_bytes = argc > 5 ? atoi(argv[5]) : 16, max_tok = argc > 6 ? atoi(argv[6]) : 8;
    const int byte_enc = argc > 7 ? atoi(argv[7]) : 1;
    const char* prefix = argc > 8 ? argv[8] : nullptr;
    int kind_err = 0;
    char err[256];
    hto_ctx* c = hto_create(argv[1], argv[2], prefix, byte_enc, &kind_err, err, sizeof err);
    if (!c) { fprintf(stderr, "%s\n", err); return 1; }
    const uint64_t seed = kind == 2 ? 0x48554732 : kind == 3 ? 0x48554733 : 0x48554735;
    std::vector<int64_t> offs(n_docs + 1);
    uint8_t* bytes = nullptr;
    const int64_t total = hutk_synth_corpus(kind, seed, 0, n_docs, 8, &bytes, offs.data());
    fprintf(stderr, "%lld bytes\n", (long long)total);
    const int64_t TILE = 960, WAVES = 4;
    const int64_t n_tiles = (total + TILE - 1) / TILE, n_wg = ((n_tiles + WAVES - 1) / WAVES + 7) / 8 * 8;
    const int64_t per_xcd = n_wg / 8;
    const int64_t RES = argc > 9 ? atoll(argv[9]) : 256;  // resident workgroups per XCD
    // words
    std::unordered_map<std::string, Info> map;
    map.reserve(1 << 22);
    struct W { int64_t wg; Info* info; uint16_t nb; };
    std::vector<W> merge_words;
    std::vector<uint32_t> starts(1 << 16);
    int64_t n_words = 0, n_single = 0, n_table = 0, n_merge = 0;
    for (int64_t d = 0; d < n_docs; d++) {
        const uint8_t* t = bytes + offs[d];
        const size_t len = (size_t)(offs[d + 1] - offs[d]);
        size_t nw = hto_split_words(t, len, starts.data(), starts.size());
        if (nw > starts.size()) { starts.resize(nw); nw = hto_split_words(t, len, starts.data(), starts.size()); }
        for (size_t w = 0; w < nw; w++) {
            const size_t s = starts[w], e = w + 1 < nw ? starts[w + 1] : len;
            const size_t nb = e - s;
            n_words++;
            if (nb == 1) { n_single++; continue; }
            std::string key((const char*)t + s, nb);
            auto it = map.find(key);
            if (it == map.end()) {
                int32_t* ids = nullptr;
                size_t n = 0;
                // a word in the middle of a document: put a letter-free separator in front so that it is not "first"
                std::string doc = std::string("\n") + key;
                hto_encode(c, (const uint8_t*)doc.data(), doc.size(), &ids, &n);
                Info inf;
                inf.ntok = (uint16_t)(n - 1);
                int units = 0;
                if (byte_enc) units = (int)nb;
                else for (size_t i = 0; i < nb; i++) units += ((uint8_t)key[i] & 0xC0) != 0x80;
                inf.nunits = (uint16_t)units;
                inf.first_round = -1;
                inf.seen_round = -1;
                inf.h = std::hash<std::string>()(key) * 0x9E3779B97F4A7C15ull;
                hto_free(ids);
                it = map.emplace(std::move(key), inf).first;
            }
            Info* inf = &it->second;
            if (inf->ntok == 1 && nb <= 14) { n_table++; continue; }
            if (inf->nunits == 1) { n_single++; continue; }
            n_merge++;
            const int64_t tile = (offs[d] + (int64_t)s) / TILE;
            merge_words.push_back({tile / WAVES, inf, (uint16_t)nb});
        }
    }
    fprintf(stderr, "words %lld single %lld table %lld merge %lld distinct %zu\n", (long long)n_words, (long long)n_single,
            (long long)n_table, (long long)n_merge, map.size());
    // rounds: workgroup g belongs to eighth g / per_xcd and to round (g % per_xcd) / RES
    auto round_of = [&](int64_t wg) { return (int32_t)((wg % per_xcd) / RES); };
    std::stable_sort(merge_words.begin(), merge_words.end(),
                     [&](const W& a, const W& b) { return round_of(a.wg) < round_of(b.wg); });
    int64_t hits = 0, inelig = 0, cold = 0, merges_all = 0, merges_left = 0, distinct_elig = 0, rejected = 0;
    // bounded, write-once table: MEMO_LOG2 slots (0 = unbounded), MEMO_CHOICES 1 or 2, MEMO_SECOND=1: a word is admitted on
    // its second sighting only (the first one leaves a mark that costs no slot)
    const int memo_log2 = getenv("MEMO_LOG2") ? atoi(getenv("MEMO_LOG2")) : 0;
    const int memo_choices = getenv("MEMO_CHOICES") ? atoi(getenv("MEMO_CHOICES")) : 2;
    const bool second_only = getenv("MEMO_SECOND") && atoi(getenv("MEMO_SECOND"));
    std::vector<uint8_t> used(memo_log2 ? (size_t)1 << memo_log2 : 1, 0);
    auto claim = [&](uint64_t h) -> bool {
        if (!memo_log2) return true;
        const uint64_t mask = ((uint64_t)1 << memo_log2) - 1;
        const uint64_t s1 = (h >> 20) & mask, s2 = ((h >> 20) ^ (((h & 0xFFFFF) | 1) * 0x5BD1u)) & mask;
        if (!used[s1]) { used[s1] = 1; return true; }
        if (memo_choices > 1 && !used[s2]) { used[s2] = 1; return true; }
        return false;
    };
    std::unordered_map<int64_t, std::pair<int, int>> wg_miss;  // wg -> {misses, longest merge count among them}
    std::unordered_map<int64_t, int> wg_max_all;
    for (const W& w : merge_words) {
        const int r = round_of(w.wg);
        const int merges = w.info->nunits - w.info->ntok;
        merges_all += merges;
        auto& ma = wg_max_all[w.wg];
        ma = std::max(ma, merges);
        const bool elig = w.nb <= key_bytes && w.info->ntok <= max_tok;
        bool hit = false;
        if (!elig) inelig++;
        else if (w.info->first_round >= 0 && w.info->first_round < r) hit = true;
        else {
            cold++;
            if (w.info->first_round < 0) {
                if (second_only && w.info->seen_round < 0) w.info->seen_round = r;
                else if (second_only && w.info->seen_round >= r) {}
                else if (w.info->first_round == -1) {
                    if (claim(w.info->h)) { w.info->first_round = r; distinct_elig++; }
                    else { w.info->first_round = -2; rejected++; }
                }
            }
        }
        if (hit) hits++;
        else {
            merges_left += merges;
            auto& m = wg_miss[w.wg];
            m.first++;
            m.second = std::max(m.second, merges);
        }
    }
    int64_t trips_all = 0, trips_left = 0, wg_with_miss = 0;
    for (auto& kv : wg_max_all) trips_all += kv.second;
    int hist[8] = {0};
    for (auto& kv : wg_miss) { trips_left += kv.second.second; wg_with_miss++; hist[std::min(kv.second.first, 7)]++; }
    printf("{\"n_docs\": %lld, \"bytes\": %lld, \"workgroups\": %lld, \"merge_words\": %lld, \"memo_hits\": %lld, "
           "\"ineligible\": %lld, \"cold_misses\": %lld, \"inserted\": %lld, \"rejected_table_full\": %lld, \"merges_all\": %lld, \"merges_left\": %lld, "
           "\"longest_per_wg_all\": %lld, \"longest_per_wg_left\": %lld, \"wg_with_miss\": %lld, "
           "\"wg_miss_hist_1_to_7plus\": [%d,%d,%d,%d,%d,%d,%d]}\n",
           (long long)n_docs, (long long)total, (long long)n_wg, (long long)n_merge, (long long)hits, (long long)inelig,
           (long long)cold, (long long)distinct_elig, (long long)rejected, (long long)merges_all, (long long)merges_left, (long long)trips_all,
           (long long)trips_left, (long long)wg_with_miss, hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7]);
    return 0;
}
