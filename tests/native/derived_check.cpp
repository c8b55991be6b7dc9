// Host check of the derived Poseidon signals (circuits_amd/csrc/derived.h, used by hz_witness_read_sym for a circom .sym of an
// unreduced compile): stdin holds lines "t hex hex ..." -- the 3 * nsbox stored S-box signals (in2, in4, out per S-box, big-endian hex)
// of one permutation -- stdout the dense trace ark.in / ark.out / mix.in / mix.out per round and lane, one hex value per line.
// tests/test_derived_signals.py compares it with the literal evaluation of circomlib's template. g++ -I circuits_amd/csrc.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "derived.h"

static void from_hex(const char* s, uint8_t* le) {
    memset(le, 0, 32);
    const size_t n = strlen(s);
    for (size_t i = 0; i < n && i < 64; i++) {
        const char c = s[n - 1 - i];
        const int v = c <= '9' ? c - '0' : (c | 32) - 'a' + 10;
        le[i / 2] |= (uint8_t)(v << (4 * (i & 1)));
    }
}
int main() {
    char* line = nullptr;
    size_t cap = 0;
    while (getline(&line, &cap, stdin) > 0) {
        std::vector<std::string> tok;
        for (char* p = strtok(line, " \n"); p; p = strtok(nullptr, " \n")) tok.push_back(p);
        if (tok.empty()) continue;
        const int t = atoi(tok[0].c_str());
        const int n = 3 * hzderived::pos_nsbox(t);
        if (t < 2 || t > 7 || (int)tok.size() != n + 1) { fprintf(stderr, "bad line: t=%d, %zu values, expected %d\n", t, tok.size() - 1, n); return 1; }
        std::vector<uint8_t> S((size_t)n * 32);
        for (int i = 0; i < n; i++) from_hex(tok[i + 1].c_str(), S.data() + 32 * i);
        std::vector<hzh::F> tr;
        hzderived::pos_trace(t, S.data(), tr);
        for (const hzh::F& v : tr) {
            uint8_t b[32];
            hzh::f_to_canon(v, b);
            for (int i = 31; i >= 0; i--) printf("%02x", b[i]);
            printf("\n");
        }
    }
    return 0;
}
