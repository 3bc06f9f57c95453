// The host-side loader (hutk_loader.cpp: vocabulary, special-character and merges files -> device table images) under
// AddressSanitizer / UBSan on the CPU.  stdin: one case per line, fields separated by TAB:
//   vocab_path  special_path  prefix|-  is_byte_encoder(0/1)  merges_path|-
// stdout: per case the load status and the sizes of the tables built (pair table, byte-pair table, word candidates:
// the loader runs its own host-side pair lookups while it builds them).
#include <cstdio>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "hutk_internal.h"

using namespace hutk;

int main() {
    std::string line;
    while (std::getline(std::cin, line)) {
        std::vector<std::string> f;
        std::stringstream ss(line);
        std::string x;
        while (std::getline(ss, x, '\t')) f.push_back(x);
        if (f.size() < 5) continue;
        Tables t;
        const LoadError e = load_tables(f[0].c_str(), f[1].c_str(), f[2] == "-" ? nullptr : f[2].c_str(), f[3] == "1",
                                        f[4] == "-" ? nullptr : f[4].c_str(), t);
        if (e.code) {
            printf("error %d\n", e.code);
            continue;
        }
        printf("ok %u symbols %lld pairs %zu pair slots %zu word candidates\n", t.n_sym, (long long)t.n_pairs, t.pair_slots.size(),
               t.cand_sym.size());
    }
    return 0;
}
