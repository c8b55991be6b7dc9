// Wire / disk formats around the witness, host code only, written against the public C ABI:
//   hz_set_inputs_json      input.json as the reference's tools write it (tools/generate-input.js:109,
//                           stringified BigInts, nested arrays) -> hz_set_input per key
//   hz_witness_write_json   witness.json, an array of decimal strings (reference tools/helpers/actions.js:136-139)
//   hz_witness_write_wtns   snarkjs .wtns: magic "wtns", version 2, section 1 = {n8 = 32, prime, nVars},
//                           section 2 = nVars x 32-byte little-endian elements (what `snarkjs wtns calculate` emits)
//   hz_symbols_write_sym    circom .sym lines "labelIdx,varIdx,componentIdx,name" for the stored signals
//   hz_symmap_*             IMPORT of a circom .sym (the compiler's own numbering of the circuit): name join against the stored
//                           signals -> the witness in circom's variable order (hz_witness_read_sym / hz_witness_write_wtns_sym),
//                           the form the reference's r1cs / zkey consume
// Together with hz_ctx_create / hz_witness_run they are the whole of the native witness binary
// `./circuit input.json witness.json` (circuits_amd/csrc/cli/hz_witness.cpp).
#include <stdio.h>
#include <string.h>
#include <new>
#include <string>
#include <vector>
#include "hostutil.h"

namespace hz {

// ---- 256-bit integers mod r, host side --------------------------------------------------------------
struct U256 { uint64_t w[5]; };   // 5th limb: headroom for x*16 + d
static const uint64_t R_LIMBS[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};

static bool ge_r(const U256& x) {
    if (x.w[4]) return true;
    for (int i = 3; i >= 0; i--) {
        if (x.w[i] > R_LIMBS[i]) return true;
        if (x.w[i] < R_LIMBS[i]) return false;
    }
    return true;
}
static void sub_r(U256& x) {
    unsigned __int128 br = 0;
    for (int i = 0; i < 5; i++) {
        const unsigned __int128 d = (unsigned __int128)x.w[i] - (i < 4 ? R_LIMBS[i] : 0) - br;
        x.w[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
}
// x <- (x * base + digit) mod r, base <= 16
static void mul_add_mod(U256& x, unsigned base, unsigned digit) {
    unsigned __int128 c = digit;
    for (int i = 0; i < 5; i++) {
        c += (unsigned __int128)x.w[i] * base;
        x.w[i] = (uint64_t)c;
        c >>= 64;
    }
    while (ge_r(x)) sub_r(x);
}
// decimal / 0x-hex / negative literal -> canonical 32 bytes (values are reduced mod r like the reference's input parser)
static bool parse_fr(const char* s, size_t n, uint8_t* out) {
    size_t i = 0;
    bool neg = false;
    if (i < n && (s[i] == '-' || s[i] == '+')) { neg = s[i] == '-'; i++; }
    unsigned base = 10;
    if (i + 1 < n && s[i] == '0' && (s[i + 1] == 'x' || s[i + 1] == 'X')) { base = 16; i += 2; }
    if (i >= n) return false;
    U256 x;
    memset(&x, 0, sizeof x);
    for (; i < n; i++) {
        unsigned d;
        const char ch = s[i];
        if (ch >= '0' && ch <= '9') d = (unsigned)(ch - '0');
        else if (base == 16 && ch >= 'a' && ch <= 'f') d = (unsigned)(ch - 'a' + 10);
        else if (base == 16 && ch >= 'A' && ch <= 'F') d = (unsigned)(ch - 'A' + 10);
        else return false;
        mul_add_mod(x, base, d);
    }
    bool zero = !(x.w[0] | x.w[1] | x.w[2] | x.w[3]);
    if (neg && !zero) {
        unsigned __int128 br = 0;
        for (int k = 0; k < 4; k++) {
            const unsigned __int128 d = (unsigned __int128)R_LIMBS[k] - x.w[k] - br;
            x.w[k] = (uint64_t)d;
            br = (d >> 64) & 1;
        }
    }
    memcpy(out, x.w, 32);
    return true;
}
// canonical 32 bytes -> decimal text
static std::string to_decimal(const uint8_t* b) {
    uint32_t w[8];
    memcpy(w, b, 32);
    char buf[80];
    int pos = 79;
    buf[pos] = 0;
    bool nz = true;
    while (nz) {
        uint64_t rem = 0;
        nz = false;
        for (int i = 7; i >= 0; i--) {
            const uint64_t cur = (rem << 32) | w[i];
            w[i] = (uint32_t)(cur / 1000000000u);
            rem = cur % 1000000000u;
            if (w[i]) nz = true;
        }
        for (int k = 0; k < 9; k++) {
            buf[--pos] = (char)('0' + rem % 10);
            rem /= 10;
            if (!nz && rem == 0) break;
        }
    }
    return std::string(buf + pos);
}

// ---- a JSON reader for input files: object of name -> (number | string | nested arrays of those) ----------
struct JsonIn {
    const char* p;
    const char* e;
    std::string err;
    int depth = 0;   // nesting of arrays: a signal has at most a few dimensions; hostile input must not overflow the stack
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
    bool fail(const char* m) { if (err.empty()) err = m; return false; }
    bool string(std::string& out) {
        if (p >= e || *p != '"') return fail("expected a string");
        p++;
        out.clear();
        while (p < e && *p != '"') {
            if (*p == '\\' && p + 1 < e) p++;
            out.push_back(*p++);
        }
        if (p >= e) return fail("unterminated string");
        p++;
        return true;
    }
    // scalar or nested array, flattened row-major into `vals`
    bool value(std::vector<uint8_t>& vals) {
        ws();
        if (p >= e) return fail("unexpected end of input");
        if (*p == '[') {
            if (depth >= 16) return fail("arrays nested deeper than 16 levels");
            p++;
            ws();
            if (p < e && *p == ']') { p++; return true; }
            depth++;
            for (;;) {
                if (!value(vals)) return false;
                ws();
                if (p < e && *p == ',') { p++; continue; }
                if (p < e && *p == ']') { p++; depth--; return true; }
                return fail("expected , or ] in array");
            }
        }
        std::string tok;
        if (*p == '"') {
            if (!string(tok)) return false;
        } else if (e - p >= 4 && !memcmp(p, "true", 4)) { tok = "1"; p += 4;
        } else if (e - p >= 5 && !memcmp(p, "false", 5)) { tok = "0"; p += 5;
        } else {
            const char* s = p;
            while (p < e && (*p == '-' || *p == '+' || (*p >= '0' && *p <= '9') || *p == 'x' || *p == 'X' || (*p >= 'a' && *p <= 'f') || (*p >= 'A' && *p <= 'F'))) p++;
            tok.assign(s, p);
        }
        if (!tok.empty() && tok.back() == 'n') tok.pop_back();   // BigInt literal suffix
        uint8_t fr[32];
        if (!parse_fr(tok.data(), tok.size(), fr)) return fail("value is not an integer");
        vals.insert(vals.end(), fr, fr + 32);
        return true;
    }
};

}  // namespace hz
using namespace hz;

extern "C" hz_status hz_set_inputs_json(hz_ctx* ctx, int32_t instance, const char* json, size_t len) {
    if (!ctx || !json) return set_err(HZ_ERR_ARG, "hz_set_inputs_json: null argument");
    JsonIn in{json, json + len, {}};
    in.ws();
    if (in.p >= in.e || *in.p != '{') return set_err(HZ_ERR_INPUT, "input JSON: expected an object");
    in.p++;
    in.ws();
    if (in.p < in.e && *in.p == '}') return HZ_OK;
    std::vector<uint8_t> vals;
    for (;;) {
        std::string key;
        in.ws();
        if (!in.string(key)) return set_err(HZ_ERR_INPUT, "input JSON: %s", in.err.c_str());
        in.ws();
        if (in.p >= in.e || *in.p != ':') return set_err(HZ_ERR_INPUT, "input JSON: expected : after \"%s\"", key.c_str());
        in.p++;
        vals.clear();
        if (!in.value(vals)) return set_err(HZ_ERR_INPUT, "input JSON, signal %s: %s", key.c_str(), in.err.c_str());
        const hz_status st = hz_set_input(ctx, instance, key.c_str(), vals.data(), vals.size() / 32);
        if (st != HZ_OK) return st;
        in.ws();
        if (in.p < in.e && *in.p == ',') { in.p++; continue; }
        if (in.p < in.e && *in.p == '}') return HZ_OK;
        return set_err(HZ_ERR_INPUT, "input JSON: expected , or } after signal %s", key.c_str());
    }
}

static const size_t CHUNK = 1 << 16;   // elements per device read

extern "C" hz_status hz_witness_write_wtns(hz_ctx* ctx, int32_t instance, const char* path) {
    if (!ctx || !path) return set_err(HZ_ERR_ARG, "hz_witness_write_wtns: null argument");
    const uint64_t n = hz_witness_len(ctx);
    if (n > 0xFFFFFFFFull) return set_err(HZ_ERR_ARG, "hz_witness_write_wtns: %llu variables do not fit the format's 32-bit count", (unsigned long long)n);
    FILE* f = fopen(path, "wb");
    if (!f) return set_err(HZ_ERR_ARG, "cannot open %s for writing", path);
    auto u32 = [&](uint32_t v) { fwrite(&v, 4, 1, f); };
    auto u64 = [&](uint64_t v) { fwrite(&v, 8, 1, f); };
    fwrite("wtns", 1, 4, f);
    u32(2);            // version
    u32(2);            // sections
    u32(1); u64(4 + 32 + 4);
    u32(32);
    fwrite(R_LIMBS, 8, 4, f);
    u32((uint32_t)n);
    u32(2); u64(n * 32);
    std::vector<uint8_t> buf(CHUNK * 32);
    for (uint64_t i = 0; i < n; i += CHUNK) {
        const uint64_t c = n - i < CHUNK ? n - i : CHUNK;
        const hz_status st = hz_witness_read(ctx, instance, i, c, buf.data());
        if (st != HZ_OK) { fclose(f); return st; }
        if (fwrite(buf.data(), 32, c, f) != c) { fclose(f); return set_err(HZ_ERR_ARG, "short write to %s", path); }
    }
    if (ferror(f)) { fclose(f); return set_err(HZ_ERR_ARG, "write to %s failed", path); }
    if (fclose(f) != 0) return set_err(HZ_ERR_ARG, "close of %s failed", path);
    return HZ_OK;
}

extern "C" hz_status hz_witness_write_json(hz_ctx* ctx, int32_t instance, const char* path) {
    if (!ctx || !path) return set_err(HZ_ERR_ARG, "hz_witness_write_json: null argument");
    const uint64_t n = hz_witness_len(ctx);
    FILE* f = fopen(path, "wb");
    if (!f) return set_err(HZ_ERR_ARG, "cannot open %s for writing", path);
    std::vector<uint8_t> buf(CHUNK * 32);
    std::string line;
    fputs("[\n", f);
    for (uint64_t i = 0; i < n; i += CHUNK) {
        const uint64_t c = n - i < CHUNK ? n - i : CHUNK;
        const hz_status st = hz_witness_read(ctx, instance, i, c, buf.data());
        if (st != HZ_OK) { fclose(f); return st; }
        line.clear();
        for (uint64_t k = 0; k < c; k++) {
            line += " \"";
            line += to_decimal(buf.data() + 32 * k);
            line += (i + k + 1 < n) ? "\",\n" : "\"\n";
        }
        fwrite(line.data(), 1, line.size(), f);
    }
    fputs("]\n", f);
    if (ferror(f)) { fclose(f); return set_err(HZ_ERR_ARG, "write to %s failed", path); }
    if (fclose(f) != 0) return set_err(HZ_ERR_ARG, "close of %s failed", path);
    return HZ_OK;
}

extern "C" hz_status hz_symbols_write_sym(const hz_ctx* ctx, const char* path) {
    if (!ctx || !path) return set_err(HZ_ERR_ARG, "hz_symbols_write_sym: null argument");
    FILE* f = fopen(path, "wb");
    if (!f) return set_err(HZ_ERR_ARG, "cannot open %s for writing", path);
    const uint64_t n = hz_symbol_count(ctx);
    for (uint64_t i = 0; i < n; i++) {
        hz_symbol s;
        const hz_status st = hz_symbol_get(ctx, i, &s);
        if (st != HZ_OK) { fclose(f); return st; }
        // every stored signal is its own variable: label index = variable index; component ids are not modelled
        fprintf(f, "%llu,%llu,0,%s\n", (unsigned long long)s.index, (unsigned long long)s.index, s.name);
    }
    if (ferror(f)) { fclose(f); return set_err(HZ_ERR_ARG, "write to %s failed", path); }
    if (fclose(f) != 0) return set_err(HZ_ERR_ARG, "close of %s failed", path);
    return HZ_OK;
}

// ---- circom .sym import --------------------------------------------------------------------------------------
// circom writes one line per signal LABEL: `labelIdx,varIdx,componentIdx,dotted.name`. varIdx is the signal's position in the
// witness the r1cs / zkey refer to, or -1 when constraint reduction eliminated it; labels that circom wired together (a component
// input and the signal it was connected to) share one varIdx. This library numbers the signals it stores in its own order
// (include/hz_layout.h) and does not store linear signals a reducing compile eliminates, so its .wtns cannot be fed to a prover
// as is. The import joins the two by NAME: variable v is resolved when ANY label of v is a stored signal. Variables none of
// whose labels is stored are reported (hz_symmap_unresolved): such a circuit build keeps signals this layout drops, and its
// witness cannot be produced from this one.
struct hz_symmap {
    std::vector<uint64_t> index;          // per variable: index in this library's per-instance witness, ~0 = unresolved
    std::vector<std::string> first_label; // per unresolved variable (in variable order): one of its names
    std::vector<uint64_t> unresolved;     // variable numbers
};

extern "C" hz_status hz_symmap_create(const hz_ctx* ctx, const char* text, size_t len, hz_symmap** out) {
    if (!ctx || !text || !out) return set_err(HZ_ERR_ARG, "hz_symmap_create: null argument");
    try {
    hz_symmap* m = new hz_symmap();
    std::vector<std::string> label;   // one label per variable, kept until the variable resolves
    const char* p = text;
    const char* e = text + len;
    uint64_t line_no = 0;
    uint64_t n_lines = 1;
    for (const char* c = text; c < e; c++) n_lines += *c == '\n';
    const uint64_t var_cap = std::min<uint64_t>(n_lines, 8 * hz_witness_len(ctx) + 1024);
    while (p < e) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
        const char* le = nl ? nl : e;
        line_no++;
        // labelIdx,varIdx,componentIdx,name
        const char* q = p;
        long long f[3] = {0, 0, 0};
        bool ok = le > p;
        for (int k = 0; k < 3 && ok; k++) {
            // bounded parse: the text need not be NUL-terminated (strtoll could run past `e`), and a field is at most 18 digits
            const char* d = q;
            const bool neg = d < le && *d == '-';
            if (neg) d++;
            long long v = 0;
            int nd = 0;
            while (d < le && *d >= '0' && *d <= '9' && nd < 19) { v = v * 10 + (*d - '0'); d++; nd++; }
            ok = nd > 0 && nd < 19 && d < le && *d == ',';
            f[k] = neg ? -v : v;
            q = d + 1;
        }
        if (le > p && !(le - p == 1 && *p == '\r')) {
            if (!ok) { delete m; return set_err(HZ_ERR_INPUT, ".sym line %llu: expected labelIdx,varIdx,componentIdx,name", (unsigned long long)line_no); }
            const char* ne = le;
            while (ne > q && (ne[-1] == '\r' || ne[-1] == ' ')) ne--;
            const long long var = f[1];
            if (var >= 0) {
                // a compile of this template cannot have more variables than the file has lines (every variable has a label), nor
                // more than a few times the signals this layout stores: a stray huge index must not become a huge allocation
                if ((uint64_t)var >= var_cap) { delete m; return set_err(HZ_ERR_INPUT, ".sym line %llu: variable index %lld out of range (cap %llu)", (unsigned long long)line_no, var, (unsigned long long)var_cap); }
                if ((uint64_t)var >= m->index.size()) { m->index.resize((size_t)var + 1, ~0ull); label.resize((size_t)var + 1); }
                if (m->index[(size_t)var] == ~0ull) {
                    const std::string name(q, (size_t)(ne - q));
                    uint64_t idx = 0;
                    if (var == 0 && (name == "one" || name == "main.one")) idx = 0, m->index[0] = 0;
                    else if (hz_symbol_lookup(ctx, name.c_str(), &idx)) m->index[(size_t)var] = idx;
                    else if (label[(size_t)var].empty()) label[(size_t)var] = name;
                }
            }
        }
        p = nl ? nl + 1 : e;
    }
    if (!m->index.empty() && m->index[0] == ~0ull) m->index[0] = 0;   // variable 0 is the constant 1 whether or not the file names it
    for (size_t v = 0; v < m->index.size(); v++)
        if (m->index[v] == ~0ull) {
            m->unresolved.push_back(v);
            m->first_label.push_back(label[v].empty() ? std::string("(no label in the file)") : label[v]);
        }
    *out = m;
    return HZ_OK;
    } catch (const std::bad_alloc&) {   // nothing may unwind through the C ABI
        return set_err(HZ_ERR_INPUT, "hz_symmap_create: out of memory while reading the .sym");
    }
}
extern "C" void hz_symmap_destroy(hz_symmap* m) { delete m; }
extern "C" uint64_t hz_symmap_nvars(const hz_symmap* m) { return m ? m->index.size() : 0; }
extern "C" uint64_t hz_symmap_unresolved(const hz_symmap* m, uint64_t i, uint64_t* var, const char** name) {
    if (!m) return 0;
    if (i < m->unresolved.size()) {
        if (var) *var = m->unresolved[i];
        if (name) *name = m->first_label[i].c_str();
    }
    return m->unresolved.size();
}
static hz_status symmap_usable(const hz_symmap* m, const char* who) {
    if (!m) return set_err(HZ_ERR_ARG, "%s: null symbol map", who);
    if (!m->unresolved.empty())
        return set_err(HZ_ERR_INPUT, "%s: %zu of %zu variables of the .sym are not stored by this layout (first: variable %llu, %s)", who, m->unresolved.size(),
                       m->index.size(), (unsigned long long)m->unresolved[0], m->first_label[0].c_str());
    return HZ_OK;
}
extern "C" hz_status hz_witness_read_sym(hz_ctx* ctx, const hz_symmap* m, int32_t instance, uint64_t first, uint64_t count, uint8_t* out) {
    const hz_status st = symmap_usable(m, "hz_witness_read_sym");
    if (st != HZ_OK) return st;
    if (first > m->index.size() || count > m->index.size() - first) return set_err(HZ_ERR_ARG, "hz_witness_read_sym: range beyond the %zu variables", m->index.size());
    return hz_witness_gather(ctx, instance, m->index.data() + first, count, out);
}
extern "C" hz_status hz_witness_write_wtns_sym(hz_ctx* ctx, const hz_symmap* m, int32_t instance, const char* path) {
    if (!ctx || !path) return set_err(HZ_ERR_ARG, "hz_witness_write_wtns_sym: null argument");
    hz_status st = symmap_usable(m, "hz_witness_write_wtns_sym");
    if (st != HZ_OK) return st;
    const uint64_t n = m->index.size();
    if (n > 0xFFFFFFFFull) return set_err(HZ_ERR_ARG, "hz_witness_write_wtns_sym: %llu variables do not fit the format's 32-bit count", (unsigned long long)n);
    FILE* f = fopen(path, "wb");
    if (!f) return set_err(HZ_ERR_ARG, "cannot open %s for writing", path);
    auto u32 = [&](uint32_t v) { fwrite(&v, 4, 1, f); };
    auto u64 = [&](uint64_t v) { fwrite(&v, 8, 1, f); };
    fwrite("wtns", 1, 4, f);
    u32(2); u32(2);
    u32(1); u64(4 + 32 + 4);
    u32(32);
    fwrite(R_LIMBS, 8, 4, f);
    u32((uint32_t)n);
    u32(2); u64(n * 32);
    std::vector<uint8_t> buf(CHUNK * 32);
    for (uint64_t i = 0; i < n; i += CHUNK) {
        const uint64_t c = n - i < CHUNK ? n - i : CHUNK;
        st = hz_witness_gather(ctx, instance, m->index.data() + i, c, buf.data());
        if (st != HZ_OK) { fclose(f); return st; }
        if (fwrite(buf.data(), 32, c, f) != c) { fclose(f); return set_err(HZ_ERR_ARG, "short write to %s", path); }
    }
    if (ferror(f)) { fclose(f); return set_err(HZ_ERR_ARG, "write to %s failed", path); }
    if (fclose(f) != 0) return set_err(HZ_ERR_ARG, "close of %s failed", path);
    return HZ_OK;
}
