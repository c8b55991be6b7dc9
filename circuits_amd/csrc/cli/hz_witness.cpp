// hz_witness -- the native witness binary of the MI355X generator: the counterpart of the circom-generated
// `./circuit input.json witness.json` the reference runs at tools/helpers/actions.js:132-146 (and of
// `snarkjs wtns calculate` for the .wtns output). Plain C++ over the C ABI of include/hermez_witness.h.
//
//   hz_witness "RollupMain(2048,32,256,64)" input.json witness.wtns [--sym out.sym] [--circom-sym circuit.sym [--circom-r1cs circuit.r1cs [--no-check]]] [--map circuit.hzmap] [--device 0]
//   hz_witness path/to/main.circom input.json witness.json
//
// The first argument is `Template(params)` or a .circom file whose `component main = Template(params);` line is
// used (the reference writes exactly such files: tools/build-circuit.js, test/*.test.js). The output format follows
// the extension (.wtns binary, anything else witness.json). Exit status 1 and the reference's error text
// ("Constraint doesn't match <lhs> != <rhs>") when a constraint fails.
// Without --circom-sym the witness is in this library's own signal numbering (--sym writes the matching symbol file): good for
// name-based checks, NOT for the reference's prover. With --circom-sym <the .sym of the circom compile> the .wtns is written
// in the compiler's variable order (name join, hz_symmap_create); it fails listing the variables this layout does not store.
// With --circom-r1cs <the .r1cs of the same compile> the variables no name resolves -- the wire-through signals of a compile without
// constraint reduction -- are solved from the circuit's linear constraints (hz_symmap_create_r1cs), and (unless --no-check) every
// constraint of the .r1cs on the witness before it is written (what `snarkjs wtns check` does; exit status 1 when one fails).
// --map <file>: the resolved map is kept there -- read when the file exists (the .sym / .r1cs are then not needed), written after an
// import otherwise: the import of a full-size circuit takes minutes, the map loads in seconds.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../../include/hermez_witness.h"

static std::string slurp(const char* path, bool* ok) {
    std::string s;
    FILE* f = fopen(path, "rb");
    *ok = f != nullptr;
    if (!f) return s;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
    fclose(f);
    return s;
}

static std::string dec(const uint8_t* b) {
    uint32_t w[8];
    memcpy(w, b, 32);
    std::string out;
    bool nz = true;
    while (nz) {
        uint64_t rem = 0;
        nz = false;
        for (int i = 7; i >= 0; i--) {
            const uint64_t cur = (rem << 32) | w[i];
            w[i] = (uint32_t)(cur / 10);
            rem = cur % 10;
            if (w[i]) nz = true;
        }
        out.insert(out.begin(), (char)('0' + rem));
    }
    return out;
}

struct Tmpl { const char* name; int id; int nparams; int order[4]; };   // order: index into {nTx, nLevels, maxL1Tx, maxFeeTx}
static const Tmpl TEMPLATES[] = {
    {"RollupMain", HZ_T_ROLLUP_MAIN, 4, {0, 1, 2, 3}}, {"RollupTx", HZ_T_ROLLUP_TX, 2, {1, 3, 0, 0}}, {"DecodeTx", HZ_T_DECODE_TX, 1, {1, 0, 0, 0}},
    {"FeeTx", HZ_T_FEE_TX, 1, {1, 0, 0, 0}}, {"HashState", HZ_T_HASH_STATE, 0, {0, 0, 0, 0}}, {"Withdraw", HZ_T_WITHDRAW, 1, {1, 0, 0, 0}},
    {"HashInputs", HZ_T_HASH_INPUTS, 4, {1, 0, 2, 3}},
    {"DecodeFloat", HZ_T_DECODE_FLOAT, 0, {0, 0, 0, 0}}, {"ComputeFee", HZ_T_COMPUTE_FEE, 0, {0, 0, 0, 0}},
    {"FeeAccumulator", HZ_T_FEE_ACCUMULATOR, 1, {3, 0, 0, 0}}, {"BalanceUpdater", HZ_T_BALANCE_UPDATER, 0, {0, 0, 0, 0}},
    {"RollupTxStates", HZ_T_ROLLUP_TX_STATES, 0, {0, 0, 0, 0}}, {"RqTxVerifier", HZ_T_RQ_TX_VERIFIER, 0, {0, 0, 0, 0}},
    {"Mux256", HZ_T_MUX256, 0, {0, 0, 0, 0}}, {"BitsCompressed2AySign", HZ_T_BITS2AYSIGN, 0, {0, 0, 0, 0}}, {"AySign2Ax", HZ_T_AYSIGN2AX, 0, {0, 0, 0, 0}},
};

static bool parse_main(std::string spec, hz_params* p) {
    const size_t at = spec.find("component main");
    if (at != std::string::npos) {
        const size_t eq = spec.find('=', at);
        if (eq == std::string::npos) return false;
        spec = spec.substr(eq + 1);
    }
    size_t i = 0;
    while (i < spec.size() && (spec[i] == ' ' || spec[i] == '\t' || spec[i] == '\n')) i++;
    size_t j = i;
    while (j < spec.size() && (isalnum((unsigned char)spec[j]) || spec[j] == '_')) j++;
    const std::string name = spec.substr(i, j - i);
    const size_t lp = spec.find('(', j), rp = spec.find(')', j);
    if (lp == std::string::npos || rp == std::string::npos || rp < lp) return false;
    std::vector<int> args;
    const std::string inner = spec.substr(lp + 1, rp - lp - 1);
    const char* q = inner.c_str();
    while (*q) {
        while (*q == ' ' || *q == ',') q++;
        if (!*q) break;
        char* end;
        args.push_back((int)strtol(q, &end, 10));
        if (end == q) return false;
        q = end;
    }
    for (const Tmpl& t : TEMPLATES) {
        if (name == t.name) {
            if ((int)args.size() != t.nparams) return false;
            int v[4] = {0, 0, 0, 0};
            for (int k = 0; k < t.nparams; k++) v[t.order[k]] = args[k];
            memset(p, 0, sizeof *p);
            p->template_id = t.id; p->nTx = v[0]; p->nLevels = v[1]; p->maxL1Tx = v[2]; p->maxFeeTx = v[3];
            return true;
        }
    }
    return false;
}

static hz_status set_json_unsupported() {
    fprintf(stderr, "--circom-sym writes .wtns only\n");
    return HZ_ERR_ARG;
}

int main(int argc, char** argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s \"Template(params)\"|main.circom input.json witness.{wtns,json} [--sym out.sym] [--circom-sym circuit.sym [--circom-r1cs circuit.r1cs [--no-check]]] [--map circuit.hzmap] [--device N]\n", argv[0]);
        return 2;
    }
    const char* sym = nullptr;
    const char* circom_sym = nullptr;
    const char* circom_r1cs = nullptr;
    const char* map_path = nullptr;
    bool check = false, no_check = false;
    int device = 0;
    for (int i = 4; i < argc; i++) {
        if (!strcmp(argv[i], "--sym") && i + 1 < argc) sym = argv[++i];
        else if (!strcmp(argv[i], "--circom-sym") && i + 1 < argc) circom_sym = argv[++i];
        else if (!strcmp(argv[i], "--circom-r1cs") && i + 1 < argc) circom_r1cs = argv[++i];
        else if (!strcmp(argv[i], "--check")) check = true;
        else if (!strcmp(argv[i], "--no-check")) no_check = true;
        else if (!strcmp(argv[i], "--map") && i + 1 < argc) map_path = argv[++i];
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if ((circom_r1cs && !circom_sym) || (check && !circom_r1cs)) { fprintf(stderr, "--circom-r1cs goes with --circom-sym, --check with --circom-r1cs\n"); return 2; }
    // an .r1cs is checked unless the caller opts out: an imported numbering has only ever been exercised on files this repository made
    // itself (no circom compiler offline), and a wrongly served variable must not reach a prover
    if (circom_r1cs && !no_check) check = true;
    std::string spec = argv[1];
    bool ok;
    if (spec.find('(') == std::string::npos) {
        spec = slurp(argv[1], &ok);
        if (!ok) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    }
    hz_params p;
    if (!parse_main(spec, &p)) { fprintf(stderr, "no supported `component main = Template(params)` in %s\n", argv[1]); return 2; }
    p.device = device;
    p.n_instances = 1;
    const std::string input = slurp(argv[2], &ok);
    if (!ok) { fprintf(stderr, "cannot read %s\n", argv[2]); return 2; }
    hz_ctx* c = nullptr;
    if (hz_ctx_create(&p, &c) != HZ_OK) { fprintf(stderr, "%s\n", hz_last_error()); return 1; }
    int rc = 0;
    hz_error err;
    hz_status st = hz_set_inputs_json(c, 0, input.data(), input.size());
    if (st == HZ_OK) st = hz_witness_run(c, &err);
    if (st == HZ_ERR_CONSTRAINT) {
        fprintf(stderr, "Error: Constraint doesn't match %s != %s (%s, unit %d)\n", dec(err.lhs).c_str(), dec(err.rhs).c_str(),
                hz_constraint_name(err.constraint_id), err.unit);
        rc = 1;
    } else if (st != HZ_OK) {
        fprintf(stderr, "Error: %s\n", hz_last_error());
        rc = 1;
    } else {
        const size_t n = strlen(argv[3]);
        const bool wtns = n > 5 && !strcmp(argv[3] + n - 5, ".wtns");
        FILE* have_map = (map_path && !check) ? fopen(map_path, "rb") : nullptr;   // (--check needs the constraints: a full import)
        if (have_map) {
            fclose(have_map);
            hz_symmap* m = nullptr;
            st = hz_symmap_load(c, map_path, &m);
            if (st == HZ_OK) {
                st = wtns ? hz_witness_write_wtns_sym(c, m, 0, argv[3]) : set_json_unsupported();
                hz_symmap_destroy(m);
            }
        } else if (circom_sym) {
            const std::string st_text = slurp(circom_sym, &ok);
            hz_symmap* m = nullptr;
            if (!ok) { fprintf(stderr, "cannot read %s\n", circom_sym); hz_ctx_destroy(c); return 2; }
            if (circom_r1cs) {
                const std::string r1 = slurp(circom_r1cs, &ok);
                if (!ok) { fprintf(stderr, "cannot read %s\n", circom_r1cs); hz_ctx_destroy(c); return 2; }
                st = hz_symmap_create_r1cs(c, st_text.data(), st_text.size(), (const uint8_t*)r1.data(), r1.size(), &m);
            } else
            st = hz_symmap_create(c, st_text.data(), st_text.size(), &m);
            if (st == HZ_OK && check && hz_symmap_unresolved(m, 0, nullptr, nullptr) == 0) {
                uint64_t n_bad = 0, first[8];
                st = hz_symmap_check_r1cs(c, m, 0, &n_bad, first, 8);
                if (st == HZ_OK && n_bad) {
                    for (uint64_t i = 0; i < n_bad && i < 8; i++) fprintf(stderr, "constraint %llu of the .r1cs does not hold\n", (unsigned long long)first[i]);
                    fprintf(stderr, "Error: %llu constraints of %s do not hold on this witness\n", (unsigned long long)n_bad, circom_r1cs);
                    hz_symmap_destroy(m);
                    hz_ctx_destroy(c);
                    return 1;
                }
                if (st == HZ_OK) fprintf(stderr, "%s: every constraint holds (%llu variables solved from linear constraints)\n", circom_r1cs, (unsigned long long)hz_symmap_solved(m));
            }
            if (st == HZ_OK) {
                const char* nm = nullptr;
                uint64_t var = 0;
                const uint64_t miss = hz_symmap_unresolved(m, 0, &var, &nm);
                for (uint64_t i = 0; i < miss && i < 10; i++) {
                    hz_symmap_unresolved(m, i, &var, &nm);
                    fprintf(stderr, "not stored by this layout: variable %llu (%s)\n", (unsigned long long)var, nm);
                }
                st = wtns ? hz_witness_write_wtns_sym(c, m, 0, argv[3]) : set_json_unsupported();
                if (st == HZ_OK && map_path) st = hz_symmap_save(c, m, map_path);
                hz_symmap_destroy(m);
            }
        } else
        st = wtns ? hz_witness_write_wtns(c, 0, argv[3]) : hz_witness_write_json(c, 0, argv[3]);
        if (st == HZ_OK && sym) st = hz_symbols_write_sym(c, sym);
        if (st != HZ_OK) { fprintf(stderr, "Error: %s\n", hz_last_error()); rc = 1; }
    }
    hz_ctx_destroy(c);
    return rc;
}
