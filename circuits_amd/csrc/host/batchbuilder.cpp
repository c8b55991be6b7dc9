// Native batch builder (include/hz_host.h): the state machine of one rollup batch, the sparse Merkle trees it rewrites and the
// EdDSA signing of synthetic transactions, in C++. Counterpart of @hermeznetwork/commonjs RollupDB / BatchBuilder (not on disk;
// reference call sites test/helpers/helpers.js:46,148, tools/generate-input.js:70-107), restated from the circuit's own rules:
// selectors of src/rollup-tx-states.circom:99-313, balances of src/balance-updater.circom:56-105, leaf multiplexers and processors of
// src/rollup-tx.circom:318-591, fee transactions of src/fee-tx.circom, the public hash of src/hash-inputs.circom:117-184, and the
// insert / update rules of circomlib's SMT (key bits LSB first, a leaf at the shallowest level where it is alone).
//
// No hash is computed while a batch is walked. Trees are pointer structures of immutable node versions; a node's hash is either a
// value or the number of a pending Poseidon JOB. Every node a job depends on lies deeper in a tree, so the dependency depth of a
// whole batch is bounded by the tree depth: the jobs are sorted into (wave, width) segments and evaluated segment by segment -- by
// hz_poseidon_dag on the device (one launch per tree level for ALL transactions) or by the host Poseidon of this library.
// Caller-side code: it prepares circuit INPUTS, never a witness, and shares nothing with oracle/.
#include <atomic>
#include <stdint.h>
#include <string.h>
#include <chrono>
#include <deque>
#include <future>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../../include/hz_host.h"
#include "bigint.h"

using hzh::U256;
using namespace hzh;

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }
struct Reject { std::string msg; };   // a transaction the circuit would reject (the Python builder's ValueError)

}  // namespace

namespace hzfee {
#define HZ_CONST_ARR static const
#include "../gen/fee_table.inc"
#undef HZ_CONST_ARR
}  // namespace hzfee

namespace {

const uint64_t P_WORDS[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
const uint64_t SUBORDER_WORDS[4] = {0x677297dc392126f1ull, 0xab3eedb83920ee0aull, 0x370a08b6d0302b0bull, 0x060c89ce5c263405ull};
const uint64_t CONST_SIG = 3322668559ull;
const uint64_t EXIT_IDX = 1;
U256 u_p() { U256 r; memcpy(r.w, P_WORDS, 32); return r; }
U256 u_suborder() { U256 r; memcpy(r.w, SUBORDER_WORDS, 32); return r; }

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- values that may still be pending ---------------------------------------------------------------------------------------------
struct Val {
    int64_t job = -1;   // >= 0: the digest of that job; job numbers run on over the flushes of a database
    U256 v = u_zero();
};
Val val_of(const U256& v) { Val r; r.v = v; return r; }

struct Job {
    uint8_t arity;
    uint32_t wave;
    int64_t in[6];   // >= 0: job number; < 0: -1 - (index of a constant)
};

// The jobs of one evaluation, detached from the database: what a worker thread evaluates while the next batch is walked
// (hzb_batch_build_begin / _finish). Jobs [base, base + jobs.size()); an input below `base` is a digest of the flush before it.
static std::atomic<long> g_live_flushes{0};   // hzb_live_flushes(): the tests' proof that a pipelined build frees what it made
struct Flush {
    Flush() { g_live_flushes++; }
    Flush(const Flush&) = delete;
    Flush& operator=(const Flush&) = delete;
    // `done` refers to the worker's shared state and the worker refers to this object by plain pointer (NOT by shared_ptr: the future inside the
    // object it keeps alive was a cycle that never died -- a few MB per pipelined batch); whoever drops the last reference waits here.
    ~Flush() { wait(); g_live_flushes--; }
    std::vector<Job> jobs;
    std::vector<U256> consts;
    int64_t base = 0;
    hzb_dag_fn fn = nullptr;
    int32_t device = 0;
    std::shared_ptr<Flush> prev;   // the flush before this one while its digests may be inputs or outputs here; cut once this one is resolved
    std::vector<U256> out;         // out[j - base] = digest of job j
    int status = HZB_OK;
    std::string err;
    uint64_t segments = 0;
    double device_ms = 0.0, eval_s = 0.0;
    std::shared_future<void> done;   // valid when a worker thread evaluates it

    void wait() const { if (done.valid()) done.wait(); }
    // the digest of job `id` (this flush's or the one before it), nullptr when neither holds it
    const U256* find(int64_t id) const {
        if (id >= base && id < base + (int64_t)out.size()) return &out[(size_t)(id - base)];
        return prev ? prev->find(id) : nullptr;
    }
    const U256& get(int64_t id) const {
        const U256* v = find(id);
        if (!v) throw Reject{"batch builder: a pending hash outlived the flush that computes it (job " + std::to_string(id) + ")"};
        return *v;
    }
    // evaluates every job: by hz_poseidon_dag on the device, segment by segment, or by the host Poseidon of this library
    void run() {
        const size_t n = jobs.size();
        out.assign(n, u_zero());
        if (prev) {
            prev->wait();
            if (prev->status != HZB_OK) { status = prev->status; err = prev->err; return; }
        }
        if (!n) return;
        const double t0 = now_s();
        if (consts.empty()) consts.push_back(u_zero());
        // digests of the flush before this one that are inputs here: constants by now
        size_t n_ext = 0;
        for (const Job& j : jobs)
            for (int k = 0; k < j.arity; k++)
                if (j.in[k] >= 0 && j.in[k] < base) n_ext++;
        // counting sort by (wave, arity)
        uint32_t max_wave = 0;
        for (const Job& j : jobs) if (j.wave > max_wave) max_wave = j.wave;
        std::vector<uint64_t> bucket((size_t)(max_wave + 1) * 8 + 1, 0);
        for (const Job& j : jobs) bucket[(size_t)j.wave * 8 + j.arity + 1]++;
        for (size_t i = 1; i < bucket.size(); i++) bucket[i] += bucket[i - 1];
        std::vector<uint32_t> order(n);
        {
            std::vector<uint64_t> pos(bucket.begin(), bucket.end() - 1);
            for (size_t i = 0; i < n; i++) order[pos[(size_t)jobs[i].wave * 8 + jobs[i].arity]++] = (uint32_t)i;
        }
        std::vector<uint32_t> seg_t;
        std::vector<uint64_t> seg_first, seg_count;
        for (size_t b = 0; b + 1 < bucket.size(); b++) {
            if (bucket[b + 1] > bucket[b]) {
                seg_t.push_back((uint32_t)(b % 8) + 1);
                seg_first.push_back(bucket[b]);
                seg_count.push_back(bucket[b + 1] - bucket[b]);
            }
        }
        const size_t n_vals = n + consts.size() + n_ext;
        if (n_vals >= (1ull << 32)) { status = HZB_ERR_ARG; err = "batch builder: more than 2^32 values in one DAG"; return; }
        std::vector<uint8_t> vals(32 * n_vals, 0);
        for (size_t c = 0; c < consts.size(); c++) u_to_bytes(consts[c], &vals[32 * (n + c)]);
        std::vector<uint32_t> job_in(6 * n), job_out(n);
        size_t ext = n + consts.size();
        for (size_t s = 0; s < n; s++) {
            const Job& j = jobs[order[s]];
            job_out[s] = order[s];
            for (int k = 0; k < 6; k++) {
                const int64_t id = j.in[k];
                if (id < 0) job_in[6 * s + k] = (uint32_t)(n + (size_t)(-1 - id));
                else if (id >= base) job_in[6 * s + k] = (uint32_t)(id - base);
                else {
                    const U256* v = k < j.arity && prev ? prev->find(id) : nullptr;
                    if (!v) { status = HZB_ERR_EVAL; err = "batch builder: a job reads a digest no flush holds (job " + std::to_string(id) + ")"; return; }
                    u_to_bytes(*v, &vals[32 * ext]);
                    job_in[6 * s + k] = (uint32_t)ext++;
                }
            }
        }
        if (fn) {
            double ms = 0.0;
            const int st = fn(device, vals.data(), n_vals, job_in.data(), job_out.data(), n, seg_t.data(), seg_first.data(), seg_count.data(), (uint32_t)seg_t.size(), &ms);
            if (st != 0) { status = HZB_ERR_EVAL; err = "batch builder: the DAG evaluator (hz_poseidon_dag) failed with status " + std::to_string(st); return; }
            device_ms += ms;
        } else {
            uint8_t buf[6 * 32];
            for (size_t s = 0; s < n; s++) {
                const Job& j = jobs[order[s]];
                for (int k = 0; k < j.arity; k++) memcpy(buf + 32 * k, &vals[32 * (size_t)job_in[6 * s + k]], 32);
                hzb_poseidon(j.arity, buf, &vals[32 * (size_t)job_out[s]]);
            }
        }
        for (size_t i = 0; i < n; i++) out[i] = u_from_bytes(&vals[32 * i]);
        segments = seg_t.size();
        eval_s = now_s() - t0;
        std::vector<Job>().swap(jobs);
        std::vector<U256>().swap(consts);
    }
};

// The pending hashes of one database: jobs, constants, and the places that wait for a digest
struct Dag {
    std::vector<Job> jobs;
    std::vector<U256> consts;
    int64_t base = 0;              // the number of jobs[0]
    std::shared_ptr<Flush> last;   // the latest flush while its digests may still be referred to by number
    hzb_dag_fn fn = nullptr;
    int32_t device = 0;
    uint64_t total_jobs = 0, total_segments = 0;
    double device_ms = 0.0, eval_s = 0.0;

    int64_t in_of(const Val& x) {
        if (x.job >= 0) return x.job;
        consts.push_back(x.v);
        return -(int64_t)consts.size();
    }
    Val poseidon(const Val* xs, int n) {
        Job j;
        j.arity = (uint8_t)n;
        j.wave = 0;
        for (int k = 0; k < 6; k++) j.in[k] = -1;
        for (int k = 0; k < n; k++) {
            j.in[k] = in_of(xs[k]);
            if (xs[k].job >= base) {
                const uint32_t w = jobs[(size_t)(xs[k].job - base)].wave;
                if (w >= j.wave) j.wave = w + 1;
            }
        }
        jobs.push_back(j);
        Val r;
        r.job = base + (int64_t)jobs.size() - 1;
        return r;
    }
    // room for a batch of `n_tx` transactions on trees of depth `levels` (two paths per transaction): no regrowth while it is walked
    void reserve_for(size_t n_tx, size_t levels) {
        const size_t n = n_tx * (2 * (levels + 2) + 8);
        if (jobs.capacity() < jobs.size() + n) jobs.reserve(jobs.size() + n);
        if (consts.capacity() < consts.size() + 2 * n) consts.reserve(consts.size() + 2 * n);
    }
    // the queued jobs as a flush of their own; the queue starts again, numbers running on
    std::shared_ptr<Flush> detach() {
        auto f = std::make_shared<Flush>();
        f->jobs.swap(jobs);
        f->consts.swap(consts);
        f->base = base;
        f->fn = fn;
        f->device = device;
        f->prev = last;
        base += (int64_t)f->jobs.size();
        last = f;
        return f;
    }
    void account(const Flush& f) {
        total_jobs += f.out.size();
        total_segments += f.segments;
        device_ms += f.device_ms;
        eval_s += f.eval_s;
    }
    // evaluates every queued job here and now; out[j] = digest of the j-th of them
    int evaluate(std::vector<U256>& out) {
        std::shared_ptr<Flush> f = detach();
        f->run();
        f->prev.reset();
        if (f->status != HZB_OK) return fail(f->status, f->err);
        account(*f);
        out = f->out;
        return HZB_OK;
    }
};

// ---- accounts ---------------------------------------------------------------------------------------------------------------------
struct Leaf {
    uint32_t token = 0;
    uint64_t nonce = 0;
    uint32_t sign = 0;
    U256 balance = u_zero(), ay = u_zero(), eth = u_zero();
};
Leaf leaf_from_c(const hzb_leaf& c) {
    Leaf l;
    l.token = c.token_id; l.nonce = c.nonce; l.sign = c.sign & 1;
    l.balance = u_from_bytes(c.balance); l.ay = u_from_bytes(c.ay); l.eth = u_from_bytes(c.eth_addr);
    return l;
}
void leaf_to_c(const Leaf& l, hzb_leaf* c) {
    memset(c, 0, sizeof(*c));
    c->token_id = l.token; c->nonce = l.nonce; c->sign = l.sign;
    u_to_bytes(l.balance, c->balance); u_to_bytes(l.ay, c->ay); u_to_bytes(l.eth, c->eth_addr);
}
// reference src/lib/hash-state.circom:14-40: Poseidon(e0 = tokenID + nonce * 2^32 + sign * 2^72, balance, ay, ethAddr)
Val hash_state(Dag& dag, const Leaf& l) {
    U256 e0 = u_from64(l.token);
    e0 = u_or(e0, u_shl(u_from64(l.nonce), 32));
    e0 = u_or(e0, u_shl(u_from64(l.sign), 72));
    const Val in[4] = {val_of(e0), val_of(l.balance), val_of(l.ay), val_of(l.eth)};
    return dag.poseidon(in, 4);
}

// ---- the pre-populated part of a state (DenseState): read-only arrays -----------------------------------------------------------------
struct Base {
    int32_t k = -1;
    uint64_t first_idx = 0, N = 0;
    std::vector<const uint8_t*> levels;
    const uint8_t* value = nullptr;
    const uint8_t* key_idx = nullptr;
    const uint64_t* mant = nullptr;
    const uint8_t* expo = nullptr;
    std::vector<Leaf> keys;   // the owner templates (token 1, nonce 0)
    bool present() const { return k >= 0; }
    bool has(uint64_t idx) const { return present() && idx >= first_idx && idx < first_idx + N; }
    uint64_t key_of(uint64_t p) const { return first_idx + ((p + N - (first_idx % N)) % N); }   // the key whose residue modulo 2^k is p
    Leaf state(uint64_t idx) const {
        const uint64_t j = idx - first_idx;
        Leaf l = keys[key_idx[j]];
        l.balance = u_mul(u_from64(mant[j]), u_pow10(expo[j]));
        return l;
    }
    U256 hash_at(int d, uint64_t p) const { return u_from_bytes(levels[d] + 32 * p); }
};

// ---- circomlib-compatible sparse Merkle tree of immutable node versions ----------------------------------------------------------------
// node ids: 0 = empty, > 0 = a node of this tree, < 0 = node (depth d, prefix p) of the base: -1 - ((d << 44) | p)
struct TNode {
    bool leaf;
    int64_t a, b;   // mid: children ids; leaf: key, index into leaf_vals
    Val h;
};
struct Found {
    bool found = false, is_old0 = false;
    std::vector<int64_t> sib;
    int64_t node = 0;       // the leaf the walk ended on (found or not)
    uint64_t leaf_key = 0;
    Val leaf_value;
};
struct SmtResult {
    bool is_old0 = false;
    uint64_t old_key = 0;
    Val old_value;
    std::vector<int64_t> sib;
};
struct Tree {
    Dag* dag = nullptr;
    const Base* base = nullptr;
    std::vector<TNode> nodes;
    std::vector<Val> leaf_vals;
    int64_t root = 0;
    size_t fresh_from = 1, fresh_vals_from = 0;   // nodes / values created since the last resolve

    Tree(Dag* d, const Base* b) : dag(d), base(b && b->present() ? b : nullptr) {
        nodes.push_back(TNode{false, 0, 0, Val()});   // id 0 stands for the empty subtree
        if (base) root = base_id(0, 0);
    }
    static int64_t base_id(int d, uint64_t p) { return -1 - (int64_t)(((uint64_t)d << 44) | p); }
    Val hash_of(int64_t id) const {
        if (id == 0) return Val();
        if (id > 0) return nodes[(size_t)id].h;
        const uint64_t e = (uint64_t)(-1 - id);
        return val_of(base->hash_at((int)(e >> 44), e & ((1ull << 44) - 1)));
    }
    Val root_hash() const { return hash_of(root); }
    // a node as (leaf?, a, b, value)
    void open(int64_t id, bool& leaf, int64_t& a, int64_t& b, Val& value) const {
        if (id > 0) {
            const TNode& n = nodes[(size_t)id];
            leaf = n.leaf; a = n.a; b = n.b;
            if (leaf) value = leaf_vals[(size_t)n.b];
            return;
        }
        const uint64_t e = (uint64_t)(-1 - id);
        const int d = (int)(e >> 44);
        const uint64_t p = e & ((1ull << 44) - 1);
        if (d == base->k) {
            leaf = true;
            a = (int64_t)base->key_of(p);
            value = val_of(u_from_bytes(base->value + 32 * ((uint64_t)a - base->first_idx)));
        } else {
            leaf = false;
            a = base_id(d + 1, p);
            b = base_id(d + 1, p + (1ull << d));
        }
    }
    Found find(uint64_t key) const {
        Found f;
        f.sib.reserve(48);
        int64_t node = root;
        int lvl = 0;
        for (;;) {
            if (node == 0) { f.is_old0 = true; return f; }
            bool leaf = false;
            int64_t a = 0, b = 0;
            Val value;
            open(node, leaf, a, b, value);
            if (leaf) {
                f.node = node;
                f.leaf_key = (uint64_t)a;
                f.leaf_value = value;
                f.found = (uint64_t)a == key;
                return f;
            }
            if ((key >> lvl) & 1) { f.sib.push_back(a); node = b; }
            else { f.sib.push_back(b); node = a; }
            lvl++;
            if (lvl > 64) throw Reject{"sparse Merkle tree deeper than 64 levels"};
        }
    }
    int64_t new_leaf(uint64_t key, const Val& value) {
        const Val in[3] = {val_of(u_from64(key)), value, val_of(u_from64(1))};
        leaf_vals.push_back(value);
        nodes.push_back(TNode{true, (int64_t)key, (int64_t)leaf_vals.size() - 1, dag->poseidon(in, 3)});
        return (int64_t)nodes.size() - 1;
    }
    int64_t up(uint64_t key, int64_t leaf, const std::vector<int64_t>& sib) {
        int64_t rt = leaf;
        for (int i = (int)sib.size() - 1; i >= 0; i--) {
            const int64_t l = ((key >> i) & 1) ? sib[(size_t)i] : rt, r = ((key >> i) & 1) ? rt : sib[(size_t)i];
            const Val in[2] = {hash_of(l), hash_of(r)};
            nodes.emplace_back(TNode{false, l, r, dag->poseidon(in, 2)});
            rt = (int64_t)nodes.size() - 1;
        }
        return rt;
    }
    SmtResult insert(uint64_t key, const Val& value) {
        Found f = find(key);
        if (f.found) throw Reject{"key exists"};
        SmtResult res;
        res.is_old0 = f.is_old0;
        res.old_key = f.is_old0 ? key : f.leaf_key;
        res.old_value = f.leaf_value;
        std::vector<int64_t> full = std::move(f.sib);
        if (!f.is_old0) {
            size_t i = full.size();
            while (((f.leaf_key >> i) & 1) == ((key >> i) & 1)) {
                full.push_back(0);
                i++;
                if (i > 64) throw Reject{"sparse Merkle tree deeper than 64 levels"};
            }
            full.push_back(f.node);   // the leaf found on the way becomes the sibling at the first level where the keys differ
        }
        const int64_t lf = new_leaf(key, value);
        root = up(key, lf, full);
        if (!f.is_old0) full.pop_back();
        while (!full.empty() && full.back() == 0) full.pop_back();
        res.sib = std::move(full);
        return res;
    }
    SmtResult update(uint64_t key, const Val& value) {
        Found f = find(key);
        if (!f.found) throw Reject{"key not found"};
        SmtResult res;
        res.old_key = key;
        res.old_value = f.leaf_value;
        const int64_t lf = new_leaf(key, value);
        root = up(key, lf, f.sib);
        res.sib = std::move(f.sib);
        return res;
    }
    // the digests of flush f into the nodes [fresh_from, nodes_end) and values [fresh_vals_from, vals_end): what existed when f was detached
    void resolve(const Flush& f, size_t nodes_end, size_t vals_end) {
        for (size_t i = fresh_from; i < nodes_end; i++) {
            Val& h = nodes[i].h;
            if (h.job >= 0) { h.v = f.get(h.job); h.job = -1; }
        }
        for (size_t i = fresh_vals_from; i < vals_end; i++) {
            Val& h = leaf_vals[i];
            if (h.job >= 0) { h.v = f.get(h.job); h.job = -1; }
        }
        if (nodes_end > fresh_from) fresh_from = nodes_end;
        if (vals_end > fresh_vals_from) fresh_vals_from = vals_end;
    }
    void resolve(const Flush& f) { resolve(f, nodes.size(), leaf_vals.size()); }
};

// ---- BabyJubjub fixed-base arithmetic and signing -----------------------------------------------------------------------------------
struct Pt { U256 x, y; };
// k * Base8 (fixed-base windows, hostlib.cpp)
Pt base8_mul(const U256& k) {
    uint8_t kk[32], ox[32], oy[32];
    u_to_bytes(k, kk);
    hzb_bjj_mul_base8(kk, ox, oy);
    return Pt{u_from_bytes(ox), u_from_bytes(oy)};
}
struct Signer {
    U256 k;
    Pt a;
};
// deterministic nonce of circuits_amd/builder.py Account.sign_msg: SHA-512(k || msg) as a little-endian integer mod the subgroup order, or 1
U256 sign_nonce(const U256& k, const U256& msg) {
    uint8_t buf[64], d[64];
    u_to_bytes(k, buf);
    u_to_bytes(msg, buf + 32);
    sha512(buf, 64, d);
    uint64_t w[8];
    memcpy(w, d, 64);
    U256 r = u_mod_wide(w, 8, u_suborder());
    if (u_is_zero(r)) r = u_from64(1);
    return r;
}
// s = (r + 8 * hm * k) mod l
U256 sign_s(const U256& r, const U256& hm, const U256& k) {
    uint64_t wide[9];
    u_mul_wide(hm, k, wide);
    wide[8] = 0;
    // times 8, plus r (hm < 2^254, k < 2^251: the product is below 2^505, times 8 below 2^508)
    uint64_t carry = 0;
    for (int i = 0; i < 8; i++) { const uint64_t nv = (wide[i] << 3) | carry; carry = wide[i] >> 61; wide[i] = nv; }
    wide[8] = carry;
    u128 c = 0;
    for (int i = 0; i < 9; i++) { c += (u128)wide[i] + (i < 4 ? r.w[i] : 0); wide[i] = (uint64_t)c; c >>= 64; }
    return u_mod_wide(wide, 9, u_suborder());
}

}  // namespace

// ---- the database and the batch ------------------------------------------------------------------------------------------------------
struct hzb_db {
    uint32_t chain_id = 1;
    uint64_t last_idx = 255;
    uint32_t num_batch = 0;
    // A batch walks and updates this state in place (tree versions, leaves, last_idx, pending hash jobs). When the walk stops half way --
    // a transaction the circuit would reject, memory -- what is left is neither the old state nor the new one: the database refuses
    // further work instead of building on it (callers that may meet rejections build on hzb_db_clone copies, as the reference's suites do).
    bool poisoned = false;
    Dag dag;
    Base base;
    Tree* state = nullptr;
    std::unordered_map<uint64_t, Leaf> leaves;
    std::unordered_map<std::string, Signer> signers;   // by private scalar
    ~hzb_db();

    bool has_leaf(uint64_t idx) const { return leaves.count(idx) || base.has(idx); }
    Leaf leaf(uint64_t idx) const {
        auto it = leaves.find(idx);
        if (it != leaves.end()) return it->second;
        return base.state(idx);
    }
    // batches whose walk is done and whose hashes are on the way (hzb_batch_build_begin), oldest first. At most one stays outstanding
    // while the next is walked: its nodes hold job numbers, the next walk's jobs take them as inputs.
    std::deque<hzb_batch*> outstanding;
    int drain();   // finishes every outstanding batch
    // the hashes queued outside a batch (direct state construction), here and now
    int flush() {
        if (int st = drain()) return st;
        if (dag.jobs.empty()) return HZB_OK;
        std::shared_ptr<Flush> f = dag.detach();
        f->run();
        if (f->status != HZB_OK) return fail(f->status, f->err);
        dag.account(*f);
        state->resolve(*f);
        f->prev.reset();
        return HZB_OK;
    }
    const Signer& signer(const U256& k) {
        const std::string key((const char*)k.w, 32);
        auto it = signers.find(key);
        if (it == signers.end()) it = signers.emplace(key, Signer{k, base8_mul(k)}).first;
        return it->second;
    }
};

namespace {

// the circuit's input signals this builder produces (src/rollup-main.circom:36-157)
#define HZB_SIGNALS(X) \
    X(oldLastIdx) X(oldStateRoot) X(globalChainID) X(currentNumBatch) X(feeIdxs) X(feePlanTokens) X(imOnChain) X(imOutIdx) X(imStateRoot) X(imExitRoot) \
    X(imAccFeeOut) X(imStateRootFee) X(imInitStateRootFee) X(imFinalAccFee) X(txCompressedData) X(amountF) X(txCompressedDataV2) X(fromIdx) X(auxFromIdx) \
    X(toIdx) X(auxToIdx) X(toBjjAy) X(toEthAddr) X(maxNumBatch) X(onChain) X(newAccount) X(rqOffset) X(rqTxCompressedDataV2) X(rqToEthAddr) X(rqToBjjAy) \
    X(s) X(r8x) X(r8y) X(loadAmountF) X(fromEthAddr) X(fromBjjCompressed) X(tokenID1) X(nonce1) X(sign1) X(balance1) X(ay1) X(ethAddr1) X(siblings1) \
    X(isOld0_1) X(oldKey1) X(oldValue1) X(tokenID2) X(nonce2) X(sign2) X(balance2) X(ay2) X(ethAddr2) X(siblings2) X(newExit) X(isOld0_2) X(oldKey2) \
    X(oldValue2) X(tokenID3) X(nonce3) X(sign3) X(balance3) X(ay3) X(ethAddr3) X(siblings3)
enum Sig {
#define X(n) S_##n,
    HZB_SIGNALS(X)
#undef X
    S_COUNT
};
const char* const SIG_NAMES[S_COUNT] = {
#define X(n) #n,
    HZB_SIGNALS(X)
#undef X
};

struct Tx {
    hzb_tx c;
    uint64_t nonce = 0;   // as signed / as written into txCompressedData
    U256 rq_v2 = u_zero(), rq_eth = u_zero(), rq_ay = u_zero();
};

U256 tx_compressed_data(const Tx& t, uint32_t chain_id) {   // src/lib/decode-tx.circom:97-141 layout
    U256 r = u_from64(CONST_SIG);
    r = u_or(r, u_shl(u_from64(chain_id), 32));
    r = u_or(r, u_shl(u_from64(t.c.from_idx), 48));
    r = u_or(r, u_shl(u_from64(t.c.to_idx), 96));
    r = u_or(r, u_shl(u_from64(t.c.token_id), 144));
    r = u_or(r, u_shl(u_from64(t.nonce), 176));
    r = u_or(r, u_shl(u_from64(t.c.user_fee), 216));
    r = u_or(r, u_shl(u_from64(t.c.to_bjj_sign & 1), 224));
    return r;
}
U256 tx_compressed_data_v2(const Tx& t) {
    U256 r = u_from64(t.c.from_idx);
    r = u_or(r, u_shl(u_from64(t.c.to_idx), 48));
    r = u_or(r, u_shl(u_from64(t.c.amount_f), 96));
    r = u_or(r, u_shl(u_from64(t.c.token_id), 136));
    r = u_or(r, u_shl(u_from64(t.nonce), 168));
    r = u_or(r, u_shl(u_from64(t.c.user_fee), 208));
    r = u_or(r, u_shl(u_from64(t.c.to_bjj_sign & 1), 216));
    return r;
}
// src/compute-fee.circom:105-143: amount * factor / 2^60 below selector 192, amount * factor from there on
U256 compute_fee(const U256& amount, unsigned sel) {
    const U256 prod = u_mul(amount, u_from64(hzfee::HZ_FEE_TABLE[sel & 255]));
    return sel < 192 ? u_shr(prod, 60) : prod;
}

}  // namespace

namespace {

// where the packed buffer waits for values
struct Out {
    uint8_t* packed = nullptr;
    uint64_t packed_bytes = 0;
    int64_t off[S_COUNT];
    uint32_t width[S_COUNT];
    struct Fix { uint64_t at; int64_t job; };
    std::vector<Fix> fixes;

    void put(Sig s, uint64_t index, const U256& v) {
        if (off[s] < 0) return;
        if (width[s] == 32) {
            const uint64_t at = (uint64_t)off[s] + 32 * index;
            if (at + 32 > packed_bytes) throw Reject{std::string("packed buffer too small for ") + SIG_NAMES[s]};
            u_to_bytes(v, packed + at);
        } else {
            const uint64_t at = (uint64_t)off[s] + index;
            if (at + 1 > packed_bytes) throw Reject{std::string("packed buffer too small for ") + SIG_NAMES[s]};
            packed[at] = (uint8_t)v.w[0];
        }
    }
    void put64(Sig s, uint64_t index, uint64_t v) { put(s, index, u_from64(v)); }
    // `count` consecutive 32-byte elements, already packed
    void put_row(Sig s, uint64_t index, const uint8_t* bytes32, uint64_t count) {
        if (off[s] < 0) return;
        if (width[s] != 32) {
            for (uint64_t k = 0; k < count; k++) put(s, index + k, u_from_bytes(bytes32 + 32 * k));
            return;
        }
        const uint64_t at = (uint64_t)off[s] + 32 * index;
        if (at + 32 * count > packed_bytes) throw Reject{std::string("packed buffer too small for ") + SIG_NAMES[s]};
        memcpy(packed + at, bytes32, 32 * count);
    }
    // bits [0, count) of v, one element each, from `index` on
    void put_bits(Sig s, uint64_t index, const U256& v, unsigned count) {
        if (off[s] < 0) return;
        if (width[s] == 1) {
            const uint64_t at = (uint64_t)off[s] + index;
            if (at + count > packed_bytes) throw Reject{std::string("packed buffer too small for ") + SIG_NAMES[s]};
            for (unsigned k = 0; k < count; k++) packed[at + k] = (uint8_t)((v.w[k >> 6] >> (k & 63)) & 1);
        } else {
            const uint64_t at = (uint64_t)off[s] + 32 * index;
            if (at + 32ull * count > packed_bytes) throw Reject{std::string("packed buffer too small for ") + SIG_NAMES[s]};
            memset(packed + at, 0, 32ull * count);
            for (unsigned k = 0; k < count; k++) packed[at + 32ull * k] = (uint8_t)((v.w[k >> 6] >> (k & 63)) & 1);
        }
    }
    void put(Sig s, uint64_t index, const Val& v) {
        if (off[s] < 0) return;
        if (v.job < 0) { put(s, index, v.v); return; }
        if (width[s] != 32) throw Reject{std::string("a hash cannot go into the narrow signal ") + SIG_NAMES[s]};
        const uint64_t at = (uint64_t)off[s] + 32 * index;
        if (at + 32 > packed_bytes) throw Reject{std::string("packed buffer too small for ") + SIG_NAMES[s]};
        fixes.push_back(Fix{at, v.job});
    }
    void resolve(const Flush& f) {
        for (const Fix& x : fixes) u_to_bytes(f.get(x.job), packed + x.at);
        std::vector<Fix>().swap(fixes);
    }
};

struct PendingSig { size_t tx; U256 msg, r; Pt r8; const Signer* signer; Val hm; };

// what the walk of a batch leaves for its second half (build_finish): the places and values that wait for the digests
struct BuildRun {
    Out o;
    uint8_t* hash_global_inputs = nullptr;
    std::shared_ptr<Flush> fl;
    std::vector<PendingSig> sigs;
    std::vector<Tx> ordered;
    std::vector<uint64_t> aux_to_v, idxs;
    uint64_t old_last_idx = 0;
    Val old_state_root, new_state_root, new_exit_root;
    size_t state_nodes_end = 0, state_vals_end = 0;
    uint64_t msg_jobs = 0, msg_segments = 0;
    double msg_device_ms = 0.0, msg_eval_s = 0.0;
    double t_start = 0.0, t_loop0 = 0.0, t_walk = 0.0;
};

}  // namespace

struct hzb_batch {
    hzb_db* db;
    int32_t nTx, L, maxL1, F;
    std::vector<Tx> txs;
    std::vector<uint32_t> fee_tokens;
    std::vector<uint64_t> fee_idxs;
    uint32_t current_num_batch;
    bool built = false;
    std::unique_ptr<BuildRun> run;   // between hzb_batch_build_begin and _finish
    Tree* exit_tree = nullptr;
    std::unordered_map<uint64_t, Leaf> exit_leaves;
    std::vector<uint8_t> nullified;
    U256 new_state_root = u_zero(), new_exit_root = u_zero();
    uint64_t new_last_idx = 0;
    uint64_t jobs = 0, segments = 0;
    double device_ms = 0.0, walk_s = 0.0, eval_s = 0.0, sign_s = 0.0;
    ~hzb_batch() { delete exit_tree; }
};

namespace {

void put_leaf(Out& o, const Sig* six, uint64_t i, const Leaf& l) {
    o.put64(six[0], i, l.token); o.put64(six[1], i, l.nonce); o.put64(six[2], i, l.sign);
    o.put(six[3], i, l.balance); o.put(six[4], i, l.ay); o.put(six[5], i, l.eth);
}
void put_siblings(Out& o, Sig s, uint64_t i, int L, const Tree& tree, const std::vector<int64_t>& sib) {
    if ((int)sib.size() > L + 1) throw Reject{"a Merkle path is longer than nLevels + 1"};
    for (int k = 0; k <= L; k++) {
        if (k < (int)sib.size()) o.put(s, i * (uint64_t)(L + 1) + (uint64_t)k, tree.hash_of(sib[(size_t)k]));
        else o.put64(s, i * (uint64_t)(L + 1) + (uint64_t)k, 0);
    }
}

int build_finish(hzb_batch* bb);

// First half of a build: the walk. Every transaction's state changes, the packed inputs that are known, the hash jobs queued; at the
// end the jobs leave as one flush -- evaluated here (threaded = false) or by a worker thread while the caller walks the next batch.
int build_begin(hzb_batch* bb, bool threaded) {
    hzb_db* db = bb->db;
    Dag& dag = db->dag;
    BuildRun& run = *bb->run;
    Out& o = run.o;
    const int L = bb->L, F = bb->F, nTx = bb->nTx;
    const double t_start = now_s();
    run.t_start = t_start;
    const U256 ETH_ANY = u_sub(u_shl(u_from64(1), 160), u_from64(1));

    int n_l1 = 0;
    for (const Tx& t : bb->txs) n_l1 += t.c.on_chain ? 1 : 0;
    if (n_l1 > bb->maxL1) throw Reject{"too many L1 txs"};
    int st = dag.jobs.empty() ? HZB_OK : db->flush();   // hashes queued outside a batch (direct state construction)
    if (st) return st;
    // one batch may be on its way while this one is walked; an older one is finished first (its digests would be out of reach)
    while (db->outstanding.size() > 1)
        if ((st = build_finish(db->outstanding.front())) != HZB_OK) return st;

    o.put64(S_oldLastIdx, 0, db->last_idx);
    run.old_state_root = db->state->root_hash();
    run.old_last_idx = db->last_idx;
    o.put(S_oldStateRoot, 0, run.old_state_root);
    o.put64(S_globalChainID, 0, db->chain_id);
    o.put64(S_currentNumBatch, 0, bb->current_num_batch);
    std::vector<uint32_t> plan(bb->fee_tokens);
    plan.resize((size_t)F, 0);
    for (int j = 0; j < F; j++) o.put64(S_feePlanTokens, (uint64_t)j, plan[(size_t)j]);
    std::vector<U256> acc_fee((size_t)F, u_zero());
    delete bb->exit_tree;
    bb->exit_tree = new Tree(&dag, nullptr);
    Tree& exit_tree = *bb->exit_tree;
    auto& exit_leaves = bb->exit_leaves;
    exit_leaves.clear();
    bb->nullified.assign((size_t)nTx, 0);

    // L1 transactions first, each group in arrival order
    std::vector<Tx>& ordered = run.ordered;
    ordered.clear();
    for (const Tx& t : bb->txs) if (t.c.on_chain) ordered.push_back(t);
    for (const Tx& t : bb->txs) if (!t.c.on_chain) ordered.push_back(t);
    // Nonces are assigned in a pre-pass: the rq fields of atomic transactions hold the neighbour's nonce
    // (src/rq-tx-verifier.circom:34-45, src/rollup-main.circom:286-309)
    {
        std::unordered_map<uint64_t, uint64_t> nonce_sim;
        for (Tx& t : ordered) {
            t.nonce = (t.c.flags & HZB_TX_HAS_NONCE) ? t.c.nonce : 0;
            if (!t.c.on_chain && t.c.from_idx) {
                const uint64_t f = t.c.from_idx;
                auto it = nonce_sim.find(f);
                const uint64_t cur = it != nonce_sim.end() ? it->second : (db->has_leaf(f) ? db->leaf(f).nonce : 0);
                if (!(t.c.flags & HZB_TX_HAS_NONCE)) t.nonce = cur;
                nonce_sim[f] = cur + 1;
            }
        }
        for (size_t i = 0; i < ordered.size(); i++) {
            Tx& t = ordered[i];
            const int k = t.c.rq_offset;
            if (t.c.flags & HZB_TX_HAS_RQ) {   // explicit rq fields (what the signer committed to) are kept as given
                t.rq_v2 = u_from_bytes(t.c.rq_tx_compressed_data_v2);
                t.rq_eth = u_from_bytes(t.c.rq_to_eth_addr);
                t.rq_ay = u_from_bytes(t.c.rq_to_bjj_ay);
            } else if (k) {
                const long j = k <= 3 ? (long)i + k : (long)i - (8 - k);
                if (j < 0 || j >= (long)ordered.size() || ordered[(size_t)j].c.on_chain)
                    throw Reject{"rqOffset " + std::to_string(k) + " of tx " + std::to_string(i) + " does not point at an L2 tx of this batch"};
                t.rq_v2 = tx_compressed_data_v2(ordered[(size_t)j]);
                t.rq_eth = u_from_bytes(ordered[(size_t)j].c.to_eth_addr);
                t.rq_ay = u_from_bytes(ordered[(size_t)j].c.to_bjj_ay);
            }
        }
    }
    // The messages of the transactions this builder signs: one Poseidon(7) each, independent of the state -- evaluated first because
    // the deterministic nonce of a signature depends on its message
    std::vector<PendingSig>& sigs = run.sigs;
    sigs.clear();
    std::vector<uint8_t> rk, rx, ry;
    const double t_sig0 = now_s();
    // (declared before r8_done: the worker below reads and writes these, and r8_done -- destroyed first -- joins it)
    Dag mdag;
    int sig_st = HZB_OK;
    struct Joined {   // (a Reject thrown by the walk must not leave the worker writing into vectors that are being unwound)
        std::future<void> f;
        ~Joined() { if (f.valid()) f.wait(); }
        Joined& operator=(std::future<void>&& g) { f = std::move(g); return *this; }
    } r8_done;
    {
        mdag.fn = dag.fn;
        mdag.device = dag.device;
        for (size_t i = 0; i < ordered.size() && i < (size_t)nTx; i++) {
            const Tx& t = ordered[i];
            if (t.c.on_chain || !(t.c.flags & HZB_TX_HAS_SIGNER) || t.c.from_idx == 0) continue;
            // src/lib/decode-tx.circom:143-160: Poseidon(txCompressedData, toEthAddr | amountF << 160 | maxNumBatch << 200, toBjjAy, rq...)
            U256 e1 = u_from_bytes(t.c.to_eth_addr);
            e1 = u_or(e1, u_shl(u_from64(t.c.amount_f), 160));
            e1 = u_or(e1, u_shl(u_from64(t.c.max_num_batch), 200));
            const Val in[6] = {val_of(tx_compressed_data(t, db->chain_id)), val_of(e1), val_of(u_from_bytes(t.c.to_bjj_ay)), val_of(t.rq_v2), val_of(t.rq_eth), val_of(t.rq_ay)};
            mdag.poseidon(in, 6);
            PendingSig ps;
            ps.tx = i;
            ps.signer = &db->signer(u_from_bytes(t.c.signer_key));
            sigs.push_back(ps);
        }
        // The walk below needs NONE of this -- not the messages, not the nonces, not R8 (the points go into the packed inputs and into
        // the hm = Poseidon(R8, A, msg) jobs after the walk; those digests are only read by build_finish, for S): the evaluation of the
        // messages on the device, the deterministic nonces and R8 = r * Base8 of every signature at once (hostlib.cpp: the additions on
        // the host's threads, one inversion for all) run on a worker BESIDE the walk. Round 5 had only R8 there and the walk waited 1.9 ms
        // per batch for the messages' round trip.
        rk.resize(32 * sigs.size()); rx.resize(32 * sigs.size()); ry.resize(32 * sigs.size());
        if (!sigs.empty()) {
            uint8_t *pk = rk.data(), *px = rx.data(), *py = ry.data();
            std::vector<PendingSig>* psigs = &sigs;
            Dag* pm = &mdag;
            int* pst = &sig_st;
            r8_done = std::async(std::launch::async, [psigs, pm, pst, pk, px, py] {
                std::vector<U256> msgs;
                *pst = pm->evaluate(msgs);
                if (*pst) return;
                std::vector<PendingSig>& sg = *psigs;
                for (size_t q = 0; q < sg.size(); q++) {
                    sg[q].msg = msgs[q];
                    sg[q].r = sign_nonce(sg[q].signer->k, sg[q].msg);
                    u_to_bytes(sg[q].r, &pk[32 * q]);
                }
                hzb_bjj_mul_base8_many(sg.size(), pk, px, py, 0);
            });
        }
    }
    bb->sign_s = now_s() - t_sig0;   // what the walk waits for: naming the message jobs; + the wait for the worker (messages, nonces, R8) below
    std::vector<int> sig_of((size_t)nTx, -1);
    for (size_t q = 0; q < sigs.size(); q++) sig_of[sigs[q].tx] = (int)q;

    static const Sig LEAF1[6] = {S_tokenID1, S_nonce1, S_sign1, S_balance1, S_ay1, S_ethAddr1};
    static const Sig LEAF2[6] = {S_tokenID2, S_nonce2, S_sign2, S_balance2, S_ay2, S_ethAddr2};
    static const Sig LEAF3[6] = {S_tokenID3, S_nonce3, S_sign3, S_balance3, S_ay3, S_ethAddr3};
    Tx nop_tx;
    memset(&nop_tx.c, 0, sizeof(nop_tx.c));
    std::vector<uint64_t>& aux_to_v = run.aux_to_v;
    aux_to_v.assign((size_t)nTx, 0);

    run.t_loop0 = now_s();
    dag.reserve_for((size_t)nTx, (size_t)L);
    db->state->nodes.reserve(db->state->nodes.size() + (size_t)nTx * 2 * ((size_t)L + 2));
    db->state->leaf_vals.reserve(db->state->leaf_vals.size() + (size_t)nTx * 2);
    o.fixes.reserve((size_t)nTx * 2 * ((size_t)L + 4));
    // imAccFeeOut[i][*] = the accumulated fees after transaction i: F values of which at most one changed -- kept as packed bytes
    std::vector<uint8_t> acc_row((size_t)F * 32, 0);
    for (int i = 0; i < nTx; i++) {
        Tx& tx = (size_t)i < ordered.size() ? ordered[(size_t)i] : nop_tx;
        const bool on = tx.c.on_chain != 0;
        const uint64_t from_idx = tx.c.from_idx, to_idx = tx.c.to_idx;
        const U256 amount = float40_to_fix(tx.c.amount_f);
        const bool has_amount = !u_is_zero(amount);
        const U256 load_amount = float40_to_fix(tx.c.load_amount_f);
        const uint32_t token = tx.c.token_id;
        const bool new_account = on && from_idx == 0;
        uint64_t aux_from = 0;
        uint64_t aux_to = (tx.c.flags & HZB_TX_HAS_AUX_TO) ? tx.c.aux_to_idx : 0;
        const U256 to_eth = u_from_bytes(tx.c.to_eth_addr), to_ay = u_from_bytes(tx.c.to_bjj_ay);
        if (!on && to_idx == 0 && from_idx && !(tx.c.flags & HZB_TX_HAS_AUX_TO)) {
            // transfer to an address / a Bjj key: the coordinator looks the receiver up -- the lowest idx with that address (and key, when
            // the address is the "any" address) holding the token
            uint64_t best = 0;
            auto consider = [&](uint64_t idx, const Leaf& lf) {
                if (lf.token != token) return;
                const bool any = u_eq(to_eth, ETH_ANY);
                const bool hit = (!any && u_eq(lf.eth, to_eth)) || (any && u_eq(lf.ay, to_ay) && lf.sign == (uint32_t)(tx.c.to_bjj_sign & 1));
                if (hit && (best == 0 || idx < best)) best = idx;
            };
            for (const auto& kv : db->leaves) consider(kv.first, kv.second);
            if (db->base.present())
                for (uint64_t idx = db->base.first_idx; idx < db->base.first_idx + db->base.N; idx++)
                    if (!db->leaves.count(idx) && (best == 0 || idx < best)) consider(idx, db->base.state(idx));
            aux_to = best;
        }
        Leaf st1, st2;
        std::vector<int64_t> sib1, sib2;
        const Tree* sib2_tree = db->state;
        uint64_t isold1 = 0, isold2 = 0, oldk1 = 0, oldk2 = 0;
        Val oldv1, oldv2;
        uint64_t new_exit = 0;
        U256 sig_r8x = u_zero(), sig_r8y = u_zero(), sig_s = u_zero();
        const U256 bjj = u_from_bytes(tx.c.from_bjj_compressed);
        const U256 from_eth = u_from_bytes(tx.c.from_eth_addr);
        if (!on && (tx.c.load_amount_f || new_account)) throw Reject{"loadAmount / newAccount on an L2 tx (the circuit rejects this tx)"};
        if (new_account) {
            db->last_idx += 1;
            aux_from = db->last_idx;
        }
        const uint64_t final_from = new_account ? aux_from : from_idx;
        const uint64_t final_to = (!on && to_idx == 0) ? aux_to : to_idx;
        const bool is_exit = final_to == EXIT_IDX;
        const bool nop = final_from == 0;
        bool is_nullified = false;
        if (!nop) {
            // ---- sender leaf as processor 1 sees it
            Leaf old1;
            if (new_account) {
                old1.token = token; old1.nonce = 0; old1.sign = u_bit(bjj, 255); old1.balance = u_zero();
                // the circuit reads the key as a FIELD element: Bits2Num over the low 254 bits (src/lib/utils-bjj.circom:22-27), i.e. mod r --
                // an invalid key such as 2^256 - 1 (reference test/rollup-main-L1.test.js:113-119) has 2^254 - 1 > r there
                old1.ay = u_low_bits(bjj, 254);
                while (u_cmp(old1.ay, u_p()) >= 0) old1.ay = u_sub(old1.ay, u_p());
                old1.eth = from_eth;
            } else {
                if (!db->has_leaf(from_idx)) throw Reject{"sender account " + std::to_string(from_idx) + " does not exist"};
                old1 = db->leaf(from_idx);
            }
            st1 = old1;
            if (!on && token != old1.token) throw Reject{"L2 tokenID does not match the sender leaf (the circuit rejects this tx)"};
            // ---- processor 2 key (src/rollup-tx-states.circom:213-221)
            const uint64_t key2 = is_exit ? final_from : (has_amount ? final_to : 0);
            const bool p2_insert = is_exit && !exit_leaves.count(key2);
            // ---- nullifiers (L1 only)
            const bool not_create = on && !new_account;
            const bool null_eth = not_create && has_amount && !u_eq(from_eth, old1.eth);
            const bool null_tok1 = not_create && token != old1.token;
            const bool null_load = null_tok1 && !u_is_zero(load_amount);
            // ---- balances
            const U256 fee = on ? u_zero() : compute_fee(amount, tx.c.user_fee);
            const U256 eff_load = (null_load || !on) ? u_zero() : load_amount;
            // the leaf processor 2 works on; a state-tree receiver is read after processor 1 has written the sender
            auto receiver_leaf = [&](const Leaf& sender_now, Leaf& out) -> bool {
                if (is_exit) {
                    auto it = exit_leaves.find(key2);
                    if (it == exit_leaves.end()) return false;
                    out = it->second;
                    return true;
                }
                if (key2 == final_from) { out = sender_now; return true; }
                if (!db->has_leaf(key2)) throw Reject{"receiver account " + std::to_string(key2) + " does not exist"};
                out = db->leaf(key2);
                return true;
            };
            // tokenID2 does not depend on balances: probe the leaf before processor 1 runs
            Leaf probe;
            const bool have_probe = has_amount ? receiver_leaf(old1, probe) : false;
            const bool null_tok2 = on && has_amount && !p2_insert && have_probe && token != probe.token;
            const bool null_amount = null_eth || null_tok2 || (null_tok1 && has_amount);
            const U256 eff_amount2 = null_amount ? u_zero() : amount;
            const U256 have = u_add(old1.balance, eff_load), want = u_add(eff_amount2, fee);
            const bool underflow_ok = u_cmp(have, want) >= 0;
            if (!on && !underflow_ok) throw Reject{"L2 underflow (the circuit rejects this tx)"};
            const U256 eff_amount3 = underflow_ok ? eff_amount2 : u_zero();
            is_nullified = !(!null_amount && underflow_ok);
            Leaf new1 = old1;
            new1.balance = u_sub(u_sub(u_add(old1.balance, eff_load), eff_amount3), fee);
            new1.nonce = old1.nonce + (on ? 0 : 1);
            SmtResult res;
            if (new_account) {
                res = db->state->insert(final_from, hash_state(dag, new1));
                isold1 = res.is_old0 ? 1 : 0;
                oldk1 = res.is_old0 ? 0 : res.old_key;
                oldv1 = res.is_old0 ? Val() : res.old_value;
            } else {
                res = db->state->update(final_from, hash_state(dag, new1));
            }
            db->leaves[final_from] = new1;
            sib1 = res.sib;
            if (!on)
                for (int j = 0; j < F; j++)
                    if (plan[(size_t)j] == token) { acc_fee[(size_t)j] = u_add(acc_fee[(size_t)j], fee); u_to_bytes(acc_fee[(size_t)j], &acc_row[32 * (size_t)j]); break; }
            // ---- processor 2: NOP unless the transaction carries an amount (nullified or not)
            if (has_amount) {
                if (is_exit) {
                    SmtResult r2;
                    Leaf enew;
                    if (p2_insert) {
                        new_exit = 1;
                        enew.token = old1.token; enew.nonce = 0; enew.sign = old1.sign; enew.balance = eff_amount3; enew.ay = old1.ay; enew.eth = old1.eth;
                        r2 = exit_tree.insert(key2, hash_state(dag, enew));
                        isold2 = r2.is_old0 ? 1 : 0;
                        oldk2 = r2.is_old0 ? 0 : r2.old_key;
                        oldv2 = r2.is_old0 ? Val() : r2.old_value;
                    } else {
                        st2 = exit_leaves[key2];
                        enew = st2;
                        enew.balance = u_add(enew.balance, eff_amount3);
                        r2 = exit_tree.update(key2, hash_state(dag, enew));
                    }
                    exit_leaves[key2] = enew;
                    sib2 = r2.sib;
                    sib2_tree = &exit_tree;
                } else {
                    Leaf rcur;
                    receiver_leaf(new1, rcur);
                    if (!on && to_idx == 0) {
                        // transferToEthAddr / transferToBjj (src/rollup-tx.circom:253-276): the signed receiver must match; 0xFF..FF selects
                        // transferToBjj, and still has to equal the leaf's ethAddr (Bjj-only accounts hold 0xFF..FF)
                        if (!u_eq(to_eth, rcur.eth)) throw Reject{"toEthAddr does not match the receiver leaf (the circuit rejects this tx)"};
                        if (u_eq(to_eth, ETH_ANY) && (!u_eq(to_ay, rcur.ay) || (uint32_t)(tx.c.to_bjj_sign & 1) != rcur.sign))
                            throw Reject{"toBjj does not match the receiver leaf (the circuit rejects this tx)"};
                    }
                    st2 = rcur;
                    Leaf rnew = rcur;
                    rnew.balance = u_add(rnew.balance, eff_amount3);
                    const SmtResult r2 = db->state->update(key2, hash_state(dag, rnew));
                    db->leaves[key2] = rnew;
                    sib2 = r2.sib;
                }
            } else if (!on) {
                st2.token = token;   // processor 2 is a NOP, but the L2 receiver-token check still compares tokenID2 (src/rollup-tx.circom:270-274)
            }
            if (!on) {
                if (sig_of[(size_t)i] < 0) {
                    if (tx.c.flags & HZB_TX_HAS_SIG) { sig_r8x = u_from_bytes(tx.c.r8x); sig_r8y = u_from_bytes(tx.c.r8y); sig_s = u_from_bytes(tx.c.s); }
                } else {
                    // R8 follows the walk (it is being computed beside it), s the evaluation
                }
            }
        }
        U256 txc;
        if (on) {
            txc = u_from64(CONST_SIG);
            txc = u_or(txc, u_shl(u_from64(db->chain_id), 32));
            txc = u_or(txc, u_shl(u_from64(from_idx), 48));
            txc = u_or(txc, u_shl(u_from64(to_idx), 96));
            txc = u_or(txc, u_shl(u_from64(token), 144));
        } else {
            txc = tx_compressed_data(tx, db->chain_id);
        }
        const uint64_t u = (uint64_t)i;
        o.put(S_txCompressedData, u, txc);
        o.put64(S_amountF, u, tx.c.amount_f);
        o.put(S_txCompressedDataV2, u, on ? u_zero() : tx_compressed_data_v2(tx));
        o.put64(S_fromIdx, u, from_idx); o.put64(S_auxFromIdx, u, aux_from);
        o.put64(S_toIdx, u, to_idx); o.put64(S_auxToIdx, u, aux_to);
        o.put(S_toBjjAy, u, to_ay); o.put(S_toEthAddr, u, to_eth);
        o.put64(S_maxNumBatch, u, tx.c.max_num_batch); o.put64(S_onChain, u, on ? 1 : 0); o.put64(S_newAccount, u, new_account ? 1 : 0);
        o.put64(S_rqOffset, u, tx.c.rq_offset); o.put(S_rqTxCompressedDataV2, u, tx.rq_v2);
        o.put(S_rqToEthAddr, u, tx.rq_eth); o.put(S_rqToBjjAy, u, tx.rq_ay);
        o.put(S_s, u, sig_s); o.put(S_r8x, u, sig_r8x); o.put(S_r8y, u, sig_r8y);
        o.put64(S_loadAmountF, u, tx.c.load_amount_f); o.put(S_fromEthAddr, u, from_eth);
        o.put_bits(S_fromBjjCompressed, u * 256, bjj, 256);
        put_leaf(o, LEAF1, u, st1);
        put_leaf(o, LEAF2, u, st2);
        put_siblings(o, S_siblings1, u, L, *db->state, sib1);
        put_siblings(o, S_siblings2, u, L, *sib2_tree, sib2);
        o.put64(S_isOld0_1, u, isold1); o.put64(S_oldKey1, u, oldk1); o.put(S_oldValue1, u, oldv1);
        o.put64(S_isOld0_2, u, isold2); o.put64(S_oldKey2, u, oldk2); o.put(S_oldValue2, u, oldv2);
        o.put64(S_newExit, u, new_exit);
        bb->nullified[(size_t)i] = is_nullified ? 1 : 0;
        aux_to_v[(size_t)i] = aux_to;
        if (i < nTx - 1) {
            o.put64(S_imOnChain, u, on ? 1 : 0); o.put64(S_imOutIdx, u, db->last_idx);
            o.put(S_imStateRoot, u, db->state->root_hash()); o.put(S_imExitRoot, u, exit_tree.root_hash());
            o.put_row(S_imAccFeeOut, u * (uint64_t)F, acc_row.data(), (uint64_t)F);
        }
        if ((size_t)i < ordered.size()) ordered[(size_t)i] = tx;
    }
    // fee transactions (src/fee-tx.circom, src/rollup-main.circom:393-431)
    o.put(S_imInitStateRootFee, 0, db->state->root_hash());
    for (int j = 0; j < F; j++) o.put(S_imFinalAccFee, (uint64_t)j, acc_fee[(size_t)j]);
    std::vector<uint64_t>& idxs = run.idxs;
    idxs = bb->fee_idxs;
    idxs.resize((size_t)F, 0);
    for (int j = 0; j < F; j++) {
        o.put64(S_feeIdxs, (uint64_t)j, idxs[(size_t)j]);
        Leaf st3;
        std::vector<int64_t> sib3;
        if (idxs[(size_t)j]) {
            if (!db->has_leaf(idxs[(size_t)j])) throw Reject{"fee account " + std::to_string(idxs[(size_t)j]) + " does not exist"};
            const Leaf cur = db->leaf(idxs[(size_t)j]);
            if (cur.token != plan[(size_t)j]) throw Reject{"fee idx token mismatch"};
            st3 = cur;
            Leaf nw = cur;
            nw.balance = u_add(nw.balance, acc_fee[(size_t)j]);
            const SmtResult r3 = db->state->update(idxs[(size_t)j], hash_state(dag, nw));
            db->leaves[idxs[(size_t)j]] = nw;
            sib3 = r3.sib;
        }
        put_leaf(o, LEAF3, (uint64_t)j, st3);
        put_siblings(o, S_siblings3, (uint64_t)j, L, *db->state, sib3);
        if (j < F - 1) o.put(S_imStateRootFee, (uint64_t)j, db->state->root_hash());
    }
    {
        const double t_r8 = now_s();
        if (r8_done.f.valid()) r8_done.f.get();
        if (sig_st) return sig_st;
        dag.total_jobs += mdag.total_jobs; dag.total_segments += mdag.total_segments; dag.device_ms += mdag.device_ms; dag.eval_s += mdag.eval_s;
        run.msg_jobs = mdag.total_jobs; run.msg_segments = mdag.total_segments; run.msg_device_ms = mdag.device_ms; run.msg_eval_s = mdag.eval_s;
        for (size_t q = 0; q < sigs.size(); q++) {
            PendingSig& ps = sigs[q];
            ps.r8 = Pt{u_from_bytes(&rx[32 * q]), u_from_bytes(&ry[32 * q])};
            o.put(S_r8x, (uint64_t)ps.tx, ps.r8.x); o.put(S_r8y, (uint64_t)ps.tx, ps.r8.y);
            const Val in[5] = {val_of(ps.r8.x), val_of(ps.r8.y), val_of(ps.signer->a.x), val_of(ps.signer->a.y), val_of(ps.msg)};
            ps.hm = dag.poseidon(in, 5);   // evaluated with the batch's Merkle hashes
        }
        bb->sign_s += now_s() - t_r8;
    }
    run.t_walk = now_s();
    // the state moves on from here: what this batch leaves behind is named now, valued when its hashes are in
    run.new_state_root = db->state->root_hash();
    run.new_exit_root = exit_tree.root_hash();
    bb->new_last_idx = db->last_idx;
    db->num_batch = bb->current_num_batch;
    run.state_nodes_end = db->state->nodes.size();
    run.state_vals_end = db->state->leaf_vals.size();
    // every hash of the batch: nLevels + 3 segments
    run.fl = dag.detach();
    db->outstanding.push_back(bb);
    if (threaded) {
        Flush* f = run.fl.get();   // kept alive by run.fl (and by the next flush's `prev`) until somebody has waited for it: ~Flush
        f->done = std::async(std::launch::async, [f] { f->run(); }).share();
    } else {
        run.fl->run();
    }
    return HZB_OK;
}

// Second half: the digests into the trees and the packed inputs, the S of the signatures, the roots, the public hash.
int build_finish(hzb_batch* bb) {
    hzb_db* db = bb->db;
    Dag& dag = db->dag;
    // in order: the trees are resolved flush by flush
    while (!db->outstanding.empty() && db->outstanding.front() != bb)
        if (int st = build_finish(db->outstanding.front())) return st;
    BuildRun& run = *bb->run;
    Out& o = run.o;
    Flush& fl = *run.fl;
    const int L = bb->L, F = bb->F, nTx = bb->nTx;
    const std::vector<Tx>& ordered = run.ordered;
    const std::vector<uint64_t>& aux_to_v = run.aux_to_v;
    const std::vector<uint64_t>& idxs = run.idxs;
    Tx nop_tx;
    memset(&nop_tx.c, 0, sizeof(nop_tx.c));
    const double t_f0 = now_s();
    fl.wait();
    db->outstanding.pop_front();
    if (fl.status != HZB_OK) return fail(fl.status, fl.err);
    dag.account(fl);
    const double t_f1 = now_s();
    db->state->resolve(fl, run.state_nodes_end, run.state_vals_end);
    bb->exit_tree->resolve(fl);
    o.resolve(fl);
    const double t_f2 = now_s();
    for (const PendingSig& ps : run.sigs) {
        const U256 s = sign_s(ps.r, fl.get(ps.hm.job), ps.signer->k);
        o.put(S_s, (uint64_t)ps.tx, s);
    }
    auto value = [&](const Val& v) { return v.job >= 0 ? fl.get(v.job) : v.v; };
    const U256 old_state_root = value(run.old_state_root);
    const uint64_t old_last_idx = run.old_last_idx;
    bb->new_state_root = value(run.new_state_root);
    bb->new_exit_root = value(run.new_exit_root);
    uint8_t* hash_global_inputs = run.hash_global_inputs;

    // the public hash (src/hash-inputs.circom:117-184)
    if (hash_global_inputs) {
        BitString bits;
        bits.be64(old_last_idx, 48); bits.be64(bb->new_last_idx, 48);
        bits.be(old_state_root, 256); bits.be(bb->new_state_root, 256); bits.be(bb->new_exit_root, 256);
        for (int i = 0; i < bb->maxL1; i++) {
            const Tx* t = (i < nTx && (size_t)i < ordered.size() && ordered[(size_t)i].c.on_chain) ? &ordered[(size_t)i] : nullptr;
            if (t) {
                bits.be(u_from_bytes(t->c.from_eth_addr), 160); bits.be(u_from_bytes(t->c.from_bjj_compressed), 256); bits.be64(t->c.from_idx, 48);
                bits.be64(t->c.load_amount_f, 40); bits.be64(t->c.amount_f, 40); bits.be64(t->c.token_id, 32); bits.be64(t->c.to_idx, 48);
            } else {
                bits.zeros(624);
            }
        }
        for (int i = 0; i < nTx; i++) {
            const Tx& t = (size_t)i < ordered.size() ? ordered[(size_t)i] : nop_tx;
            const bool on = t.c.on_chain != 0;
            const uint64_t final_to = (!on && t.c.to_idx == 0) ? aux_to_v[(size_t)i] : t.c.to_idx;
            bits.be64(t.c.from_idx, (unsigned)L); bits.be64(final_to, (unsigned)L);
            bits.be64(bb->nullified[(size_t)i] ? 0 : t.c.amount_f, 40);
            bits.be64(on ? 0 : t.c.user_fee, 8);
        }
        for (int j = 0; j < F; j++) bits.be64(idxs[(size_t)j], (unsigned)L);
        bits.be64(db->chain_id, 16); bits.be64(bb->current_num_batch, 32);
        uint8_t d[32];
        sha256_bits(bits, d);
        U256 h;
        for (int i = 0; i < 32; i++) ((uint8_t*)h.w)[i] = d[31 - i];
        while (u_cmp(h, u_p()) >= 0) h = u_sub(h, u_p());
        u_to_bytes(h, hash_global_inputs);
    }
    if (getenv("HZB_TIMING"))
        fprintf(stderr, "hzb build: pre-walk+sign %.2f ms, walk loop %.2f, wait for the flush %.2f (evaluator %.2f), resolve %.2f, tail (S, public hash) %.2f\n",
                (run.t_loop0 - run.t_start) * 1e3, (run.t_walk - run.t_loop0) * 1e3, (t_f1 - t_f0) * 1e3, fl.eval_s * 1e3, (t_f2 - t_f1) * 1e3, (now_s() - t_f2) * 1e3);
    bb->built = true;
    bb->jobs = fl.out.size() + run.msg_jobs;
    bb->segments = fl.segments + run.msg_segments;
    bb->device_ms = fl.device_ms + run.msg_device_ms;
    bb->eval_s = fl.eval_s + run.msg_eval_s;
    bb->walk_s = (run.t_walk - run.t_start) - run.msg_eval_s;   // the walk proper: bookkeeping, nonces and R8 of the signatures
    fl.prev.reset();   // nothing of the flush before it is needed through this one any more
    if (dag.last.get() == &fl && db->outstanding.empty()) dag.last.reset();   // every node is a value again: no number refers to it
    bb->run.reset();
    return HZB_OK;
}

}  // namespace

int hzb_db::drain() {
    while (!outstanding.empty()) {
        hzb_batch* b = outstanding.front();
        try {
            if (int st = build_finish(b)) { poisoned = true; return st; }
        } catch (const Reject& r) {
            poisoned = true;
            return fail(HZB_ERR_REJECTED, r.msg);
        }
    }
    return HZB_OK;
}
hzb_db::~hzb_db() {
    for (hzb_batch* b : outstanding)   // their worker threads read flushes the batches own: let them end
        if (b->run && b->run->fl) b->run->fl->wait();
    delete state;
}

// ---- C ABI ------------------------------------------------------------------------------------------------------------------------------
extern "C" {

const char* hzb_last_error(void) { return g_err.c_str(); }

// the working copy the reference's suites build a batch on before they consolidate it (rollupDb.buildBatch on a copy of the state)
static const char* const POISONED = "the database was left half-updated by a batch that failed to build: discard it (build on hzb_db_clone copies)";
hzb_db* hzb_db_clone(const hzb_db* src) {
    if (!src) { fail(HZB_ERR_ARG, "hzb_db_clone: null database"); return nullptr; }
    if (src->poisoned) { fail(HZB_ERR_REJECTED, POISONED); return nullptr; }
    if (const_cast<hzb_db*>(src)->drain() != HZB_OK) return nullptr;   // a batch on its way: the copy starts from values, not from job numbers
    hzb_db* db = new hzb_db();
    db->chain_id = src->chain_id; db->last_idx = src->last_idx; db->num_batch = src->num_batch;
    db->dag = src->dag;
    db->base = src->base;
    db->leaves = src->leaves;
    db->signers = src->signers;
    db->state = new Tree(*src->state);
    db->state->dag = &db->dag;
    db->state->base = db->base.present() ? &db->base : nullptr;
    return db;
}
hzb_db* hzb_db_create(uint32_t chain_id, uint64_t first_idx) {
    hzb_db* db = new hzb_db();
    db->chain_id = chain_id;
    db->last_idx = first_idx ? first_idx - 1 : 0;
    db->state = new Tree(&db->dag, nullptr);
    return db;
}
void hzb_db_destroy(hzb_db* db) { delete db; }
int hzb_db_set_dag(hzb_db* db, hzb_dag_fn fn, int32_t device) {
    if (!db) return fail(HZB_ERR_ARG, "hzb_db_set_dag: null database");
    db->dag.fn = fn;
    db->dag.device = device;
    return HZB_OK;
}
int hzb_db_set_base(hzb_db* db, int32_t k, uint64_t first_idx, const uint8_t* const* levels, const uint8_t* value, const uint8_t* key_idx, const uint64_t* mant,
                    const uint8_t* expo, int32_t n_keys, const uint8_t* key_sign, const uint8_t* key_ay, const uint8_t* key_eth) {
    if (!db || !levels || !value || !key_idx || !mant || !expo || k < 0 || k > 40 || n_keys < 1 || n_keys > 256) return fail(HZB_ERR_ARG, "hzb_db_set_base: bad arguments");
    if (db->state->nodes.size() > 1 || !db->leaves.empty() || db->base.present()) return fail(HZB_ERR_ARG, "hzb_db_set_base: the database is not empty");
    Base& b = db->base;
    b.k = k; b.first_idx = first_idx; b.N = 1ull << k;
    b.levels.assign(levels, levels + k + 1);
    b.value = value; b.key_idx = key_idx; b.mant = mant; b.expo = expo;
    for (int i = 0; i < n_keys; i++) {
        Leaf l;
        l.token = 1; l.sign = key_sign[i] & 1; l.ay = u_from_bytes(key_ay + 32 * i); l.eth = u_from_bytes(key_eth + 32 * i);
        b.keys.push_back(l);
    }
    delete db->state;
    db->state = new Tree(&db->dag, &db->base);
    db->last_idx = first_idx + b.N - 1;
    return HZB_OK;
}
int hzb_db_add_account(hzb_db* db, const hzb_leaf* leaf, uint64_t* idx) {
    if (!db || !leaf) return fail(HZB_ERR_ARG, "hzb_db_add_account: null argument");
    if (db->poisoned) return fail(HZB_ERR_REJECTED, POISONED);
    try {
        const Leaf l = leaf_from_c(*leaf);   // rejects before anything is touched
        const uint64_t at = db->last_idx + 1;
        db->leaves[at] = l;
        db->state->insert(at, hash_state(db->dag, l));
        db->last_idx = at;                   // only once the tree holds the leaf
        if (idx) *idx = at;
    } catch (const Reject& r) {
        db->leaves.erase(db->last_idx + 1);
        return fail(HZB_ERR_REJECTED, r.msg);
    } catch (const std::bad_alloc&) {
        db->poisoned = true;                 // the tree may hold half a path
        return fail(HZB_ERR_ARG, "hzb_db_add_account: out of memory");
    }
    return HZB_OK;
}
int hzb_db_get_account(hzb_db* db, uint64_t idx, hzb_leaf* out) {
    if (!db || !out) return fail(HZB_ERR_ARG, "hzb_db_get_account: null argument");
    if (!db->has_leaf(idx)) return fail(HZB_ERR_ARG, "account " + std::to_string(idx) + " does not exist");
    leaf_to_c(db->leaf(idx), out);
    return HZB_OK;
}
int hzb_db_state_root(hzb_db* db, uint8_t* out) {
    if (!db || !out) return fail(HZB_ERR_ARG, "hzb_db_state_root: null argument");
    if (db->poisoned) return fail(HZB_ERR_REJECTED, POISONED);
    const int st = db->flush();
    if (st) return st;
    u_to_bytes(db->state->root_hash().v, out);
    return HZB_OK;
}
uint64_t hzb_db_last_idx(const hzb_db* db) { return db ? db->last_idx : 0; }
uint32_t hzb_db_num_batch(const hzb_db* db) { return db ? db->num_batch : 0; }

hzb_batch* hzb_batch_create(hzb_db* db, int32_t n_tx, int32_t n_levels, int32_t max_l1, int32_t max_fee) {
    if (!db || n_tx < 1 || n_levels < 1 || n_levels > 48 || max_l1 < 0 || max_fee < 1) { fail(HZB_ERR_ARG, "hzb_batch_create: bad parameters"); return nullptr; }
    if (db->poisoned) { fail(HZB_ERR_REJECTED, POISONED); return nullptr; }
    hzb_batch* b = new hzb_batch();
    b->db = db; b->nTx = n_tx; b->L = n_levels; b->maxL1 = max_l1; b->F = max_fee;
    b->current_num_batch = db->num_batch + 1;
    return b;
}
void hzb_batch_destroy(hzb_batch* b) {
    if (!b) return;
    if (b->run && b->run->fl) {   // begun, never finished: the state's nodes must not keep job numbers nobody resolves
        bool queued = false;
        for (hzb_batch* q : b->db->outstanding) queued = queued || q == b;
        if (queued) {
            try {
                if (build_finish(b) != HZB_OK) b->db->poisoned = true;
            } catch (const Reject&) { b->db->poisoned = true; }
            for (auto it = b->db->outstanding.begin(); it != b->db->outstanding.end(); ++it)
                if (*it == b) { b->db->outstanding.erase(it); break; }
        }
        if (b->run && b->run->fl) b->run->fl->wait();
    }
    delete b;
}
int hzb_batch_add_tx(hzb_batch* b, const hzb_tx* tx) {
    if (!b || !tx) return fail(HZB_ERR_ARG, "hzb_batch_add_tx: null argument");
    if ((int)b->txs.size() >= b->nTx) return fail(HZB_ERR_REJECTED, "batch full");
    if (tx->amount_f >> 40 || tx->load_amount_f >> 40) return fail(HZB_ERR_ARG, "hzb_batch_add_tx: a float40 has 40 bits");
    Tx t;
    t.c = *tx;
    b->txs.push_back(t);
    return HZB_OK;
}
int hzb_batch_add_txs(hzb_batch* b, const hzb_tx* txs, uint64_t n) {
    if (!b || (!txs && n)) return fail(HZB_ERR_ARG, "hzb_batch_add_txs: null argument");
    for (uint64_t i = 0; i < n; i++)
        if (int st = hzb_batch_add_tx(b, txs + i)) return st;   // the ones before it stay added, as with single calls
    return 0;
}
// ---- the synthetic benchmark batch, end to end in this library ----------------------------------------------------------------------------
// The recipe of the reference's generator (tools/generate-input.js:61-109, tools/helpers/gen-inputs-utils.js:6-71): maxL1Tx
// createAccountDeposits of a random key with a random float40 load amount, then signed L2 transfers of 20 % of the sender's balance
// between random accounts of the pre-populated state (the first `exits` of them exits), userFee 176, one fee token and one fee
// receiver. circuits_amd/native_builder.py states the same recipe in Python on Python's own generator; here it runs on a bit-exact
// MT19937 + randrange, so both produce the same transactions and hence byte-identical circuit inputs (tests/test_native_builder.py).
namespace {
struct PyRandom {   // CPython's random.Random(seed) for a non-negative integer seed: init_by_array over its 32-bit words
    uint32_t mt[624];
    int idx = 624;
    explicit PyRandom(uint64_t seed) {
        uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
        const int klen = key[1] ? 2 : 1;
        mt[0] = 19650218u;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        int i = 1, j = 0;
        for (int k = 624 > klen ? 624 : klen; k; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            if (++i >= 624) { mt[0] = mt[623]; i = 1; }
            if (++j >= klen) j = 0;
        }
        for (int k = 623; k; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
            if (++i >= 624) { mt[0] = mt[623]; i = 1; }
        }
        mt[0] = 0x80000000u;
    }
    uint32_t next() {
        if (idx >= 624) {
            for (int k = 0; k < 624; k++) {
                const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    }
    U256 getrandbits(int k) {   // k <= 256: 32-bit words, least significant first, the last one shifted down
        U256 r = u_zero();
        for (int w = 0; k > 0; w++, k -= 32) {
            uint32_t x = next();
            if (k < 32) x >>= (32 - k);
            r.w[w >> 1] |= (uint64_t)x << (32 * (w & 1));
        }
        return r;
    }
    static int bit_length(const U256& n) {
        for (int i = 255; i >= 0; i--)
            if (u_bit(n, (unsigned)i)) return i + 1;
        return 0;
    }
    U256 randbelow(const U256& n) {   // random.randrange(n): rejection on getrandbits(n.bit_length())
        const int k = bit_length(n);
        U256 r = getrandbits(k);
        while (u_cmp(r, n) >= 0) r = getrandbits(k);
        return r;
    }
    uint64_t randbelow64(uint64_t n) { return randbelow(u_from64(n)).w[0]; }
};
U256 u_div_small(const U256& a, uint64_t d, uint64_t* rem = nullptr) {
    U256 q;
    unsigned __int128 r = 0;
    for (int i = 3; i >= 0; i--) {
        const unsigned __int128 cur = (r << 64) | a.w[i];
        q.w[i] = (uint64_t)(cur / d);
        r = cur % d;
    }
    if (rem) *rem = (uint64_t)r;
    return q;
}
uint64_t floor_fix2float(const U256& v) {   // the largest float40 not above v (builder.py floor_fix2float)
    if (u_is_zero(v)) return 0;
    U256 m = v;
    for (uint64_t e = 0; e < 32; e++) {
        if (!(m.w[1] | m.w[2] | m.w[3]) && m.w[0] < (1ull << 35)) return m.w[0] + (e << 35);
        m = u_div_small(m, 10);
    }
    return 0;
}
}  // namespace

int hzb_batch_add_synthetic(hzb_batch* b, uint64_t seed, int32_t exits, int32_t n_keys, const uint8_t* l1_bjj_compressed, const uint8_t* l1_eth_addr,
                            int32_t n_signers, const uint8_t* signer_keys) {
    if (!b || !l1_bjj_compressed || !l1_eth_addr || !signer_keys || n_keys < 1 || n_signers < 1) return fail(HZB_ERR_ARG, "hzb_batch_add_synthetic: null argument");
    hzb_db* db = b->db;
    if (!db->base.present() || !b->txs.empty()) return fail(HZB_ERR_ARG, "hzb_batch_add_synthetic: needs an empty batch on a database with a pre-populated state (hzb_db_set_base)");
    const Base& base = db->base;
    PyRandom rng(seed);
    const int n_tx = b->nTx, n_l1 = b->maxL1 < n_tx ? b->maxL1 : n_tx;
    const U256 two96 = u_shl(u_from64(1), 96);
    b->txs.reserve((size_t)n_tx);
    for (int i = 0; i < n_l1; i++) {
        Tx t;
        memset(&t.c, 0, sizeof t.c);
        const uint64_t q = rng.randbelow64((uint64_t)n_keys);
        t.c.load_amount_f = floor_fix2float(rng.randbelow(two96));
        t.c.on_chain = 1;
        t.c.token_id = 1;
        memcpy(t.c.from_bjj_compressed, l1_bjj_compressed + 32 * q, 32);
        memcpy(t.c.from_eth_addr, l1_eth_addr + 32 * q, 32);
        b->txs.push_back(t);
    }
    struct Acc { U256 bal; uint64_t nonce; };
    std::unordered_map<uint64_t, Acc> tmp;
    auto pick = [&]() { return base.first_idx + rng.randbelow64(base.N); };
    auto account = [&](uint64_t idx) -> Acc {
        auto it = tmp.find(idx);
        if (it != tmp.end()) return it->second;
        const Leaf l = db->leaf(idx);
        return Acc{l.balance, l.nonce};
    };
    for (int t = 0; t < n_tx - n_l1; t++) {
        const uint64_t frm = pick(), to = pick();
        const Acc a = account(frm);
        const uint64_t amount_f = floor_fix2float(u_div_small(u_mul(a.bal, u_from64(20)), 100));
        const U256 amount = float40_to_fix(amount_f);
        const bool is_exit = t < exits;
        Tx x;
        memset(&x.c, 0, sizeof x.c);
        x.c.from_idx = frm; x.c.to_idx = is_exit ? 1 : to; x.c.amount_f = amount_f; x.c.nonce = a.nonce;
        x.c.user_fee = 176; x.c.token_id = 1; x.c.flags = HZB_TX_HAS_NONCE | HZB_TX_HAS_SIGNER;
        const uint64_t ki = base.key_idx[frm - base.first_idx];
        if ((int64_t)ki >= n_signers) return fail(HZB_ERR_ARG, "hzb_batch_add_synthetic: account " + std::to_string(frm) + " belongs to key " + std::to_string(ki) + " of " + std::to_string(n_signers));
        memcpy(x.c.signer_key, signer_keys + 32 * ki, 32);
        b->txs.push_back(x);
        const U256 nb = u_sub(u_sub(a.bal, amount), compute_fee(amount, 176));
        tmp[frm] = Acc{nb, a.nonce + 1};
        if (!is_exit && to != frm) {
            const Acc r = account(to);
            tmp[to] = Acc{u_add(r.bal, amount), r.nonce};
        } else if (!is_exit) {
            tmp[frm] = Acc{u_add(nb, amount), a.nonce + 1};
        }
    }
    if (int st = hzb_batch_add_token(b, 1)) return st;
    return hzb_batch_add_fee_idx(b, pick());
}

int hzb_batch_add_token(hzb_batch* b, uint32_t token_id) {
    if (!b) return fail(HZB_ERR_ARG, "hzb_batch_add_token: null batch");
    if ((int)b->fee_tokens.size() >= b->F) return fail(HZB_ERR_REJECTED, "fee plan full");
    b->fee_tokens.push_back(token_id);
    return HZB_OK;
}
int hzb_batch_add_fee_idx(hzb_batch* b, uint64_t idx) {
    if (!b) return fail(HZB_ERR_ARG, "hzb_batch_add_fee_idx: null batch");
    if ((int)b->fee_idxs.size() >= b->F) return fail(HZB_ERR_REJECTED, "fee plan full");
    b->fee_idxs.push_back(idx);
    return HZB_OK;
}
static int batch_build_begin(hzb_batch* b, int32_t n_signals, const char* const* names, const uint64_t* offsets, const uint32_t* widths, uint8_t* packed,
                             uint64_t packed_bytes, uint8_t* hash_global_inputs, bool threaded, const char* who) {
    if (!b || !names || !offsets || !widths || !packed || n_signals < 0) return fail(HZB_ERR_ARG, std::string(who) + ": null argument");
    if (b->built || b->run) return fail(HZB_ERR_ARG, std::string(who) + ": the batch has been built");
    if (b->db->poisoned) return fail(HZB_ERR_REJECTED, POISONED);
    b->run.reset(new BuildRun());
    Out& o = b->run->o;
    o.packed = packed;
    o.packed_bytes = packed_bytes;
    b->run->hash_global_inputs = hash_global_inputs;
    for (int s = 0; s < S_COUNT; s++) { o.off[s] = -1; o.width[s] = 32; }
    for (int i = 0; i < n_signals; i++) {
        int s = 0;
        while (s < S_COUNT && strcmp(SIG_NAMES[s], names[i]) != 0) s++;
        if (s == S_COUNT) { b->run.reset(); return fail(HZB_ERR_ARG, std::string(who) + ": the batch builder does not produce the input signal " + names[i]); }
        if (widths[i] != 32 && widths[i] != 1) { b->run.reset(); return fail(HZB_ERR_ARG, std::string(who) + ": element width of " + names[i] + " must be 32 or 1"); }
        o.off[s] = (int64_t)offsets[i];
        o.width[s] = widths[i];
    }
    try {
        const int st = build_begin(b, threaded);
        if (st != HZB_OK) { b->db->poisoned = true; b->run.reset(); }
        return st;
    } catch (const Reject& r) {
        b->db->poisoned = true;
        b->run.reset();
        return fail(HZB_ERR_REJECTED, r.msg);
    } catch (const std::bad_alloc&) {
        b->db->poisoned = true;
        b->run.reset();
        return fail(HZB_ERR_ARG, std::string(who) + ": out of memory");
    }
}
static int batch_build_finish(hzb_batch* b, const char* who) {
    if (!b) return fail(HZB_ERR_ARG, std::string(who) + ": null batch");
    if (b->built) return HZB_OK;
    if (!b->run || !b->run->fl) return fail(HZB_ERR_ARG, std::string(who) + ": the batch's build has not begun");
    try {
        const int st = build_finish(b);
        if (st != HZB_OK) b->db->poisoned = true;   // the evaluator failed after the walk had updated the state
        return st;
    } catch (const Reject& r) {
        b->db->poisoned = true;
        return fail(HZB_ERR_REJECTED, r.msg);
    } catch (const std::bad_alloc&) {
        b->db->poisoned = true;
        return fail(HZB_ERR_ARG, std::string(who) + ": out of memory");
    }
}
int hzb_batch_build(hzb_batch* b, int32_t n_signals, const char* const* names, const uint64_t* offsets, const uint32_t* widths, uint8_t* packed, uint64_t packed_bytes,
                    uint8_t* hash_global_inputs) {
    if (int st = batch_build_begin(b, n_signals, names, offsets, widths, packed, packed_bytes, hash_global_inputs, false, "hzb_batch_build")) return st;
    return batch_build_finish(b, "hzb_batch_build");
}
int hzb_batch_build_begin(hzb_batch* b, int32_t n_signals, const char* const* names, const uint64_t* offsets, const uint32_t* widths, uint8_t* packed, uint64_t packed_bytes,
                          uint8_t* hash_global_inputs) {
    return batch_build_begin(b, n_signals, names, offsets, widths, packed, packed_bytes, hash_global_inputs, true, "hzb_batch_build_begin");
}
int hzb_batch_build_finish(hzb_batch* b) { return batch_build_finish(b, "hzb_batch_build_finish"); }
int hzb_batch_roots(const hzb_batch* b, uint8_t* new_state_root, uint8_t* new_exit_root, uint64_t* new_last_idx) {
    if (!b || !b->built) return fail(HZB_ERR_ARG, "hzb_batch_roots: the batch has not been built");
    if (new_state_root) u_to_bytes(b->new_state_root, new_state_root);
    if (new_exit_root) u_to_bytes(b->new_exit_root, new_exit_root);
    if (new_last_idx) *new_last_idx = b->new_last_idx;
    return HZB_OK;
}
int hzb_batch_exit_proof(hzb_batch* b, uint64_t idx, hzb_leaf* leaf, uint8_t* siblings, int32_t* n_siblings) {
    if (!b || !b->built || !leaf || !siblings) return fail(HZB_ERR_ARG, "hzb_batch_exit_proof: the batch has not been built");
    auto it = b->exit_leaves.find(idx);
    if (it == b->exit_leaves.end()) return fail(HZB_ERR_ARG, "no exit leaf for account " + std::to_string(idx));
    leaf_to_c(it->second, leaf);
    const Found f = b->exit_tree->find(idx);
    if (!f.found || (int)f.sib.size() > b->L + 1) return fail(HZB_ERR_ARG, "exit tree: leaf not found");
    memset(siblings, 0, 32 * (size_t)(b->L + 1));
    for (size_t k = 0; k < f.sib.size(); k++) u_to_bytes(b->exit_tree->hash_of(f.sib[k]).v, siblings + 32 * k);
    if (n_siblings) *n_siblings = (int32_t)f.sib.size();
    return HZB_OK;
}
int hzb_batch_tx_flags(const hzb_batch* b, int32_t i, int32_t* is_amount_nullified) {
    if (!b || !b->built || i < 0 || i >= b->nTx) return fail(HZB_ERR_ARG, "hzb_batch_tx_flags: bad arguments");
    if (is_amount_nullified) *is_amount_nullified = b->nullified[(size_t)i];
    return HZB_OK;
}
double hzb_batch_sign_s(const hzb_batch* b) { return b ? b->sign_s : 0.0; }
long hzb_live_flushes(void) { return g_live_flushes.load(); }
int hzb_batch_stats(const hzb_batch* b, uint64_t* jobs, uint64_t* segments, double* device_ms, double* walk_s, double* eval_s) {
    if (!b) return fail(HZB_ERR_ARG, "hzb_batch_stats: null batch");
    if (jobs) *jobs = b->jobs;
    if (segments) *segments = b->segments;
    if (device_ms) *device_ms = b->device_ms;
    if (walk_s) *walk_s = b->walk_s;
    if (eval_s) *eval_s = b->eval_s;
    return HZB_OK;
}

int hzb_eddsa_pubkey(const uint8_t* key, uint8_t* ax, uint8_t* ay) {
    const Pt a = base8_mul(u_from_bytes(key));
    u_to_bytes(a.x, ax);
    u_to_bytes(a.y, ay);
    return 0;
}
int hzb_eddsa_sign(const uint8_t* key, const uint8_t* msg, uint8_t* r8x, uint8_t* r8y, uint8_t* s) {
    const U256 k = u_from_bytes(key), m = u_from_bytes(msg);
    const Pt a = base8_mul(k);
    const U256 r = sign_nonce(k, m);
    const Pt r8 = base8_mul(r);
    uint8_t in[5 * 32], hm[32];
    u_to_bytes(r8.x, in); u_to_bytes(r8.y, in + 32); u_to_bytes(a.x, in + 64); u_to_bytes(a.y, in + 96); u_to_bytes(m, in + 128);
    hzb_poseidon(5, in, hm);
    u_to_bytes(r8.x, r8x);
    u_to_bytes(r8.y, r8y);
    u_to_bytes(sign_s(r, u_from_bytes(hm), k), s);
    return 0;
}

}  // extern "C"
