// hz_ctx: one "compiled circuit" -- layout, HBM-resident witness buffer, input upload, kernel
// schedule, constraint-failure reporting, symbol table. Implements the context part of
// include/hermez_witness.h (the calls that replace circom's tester()/calculateWitness()/assertOut(),
// reference test/helpers/helpers.js:139-155).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>
#include "hostutil.h"
#include "kernels.h"
#include "tx_dev.h"
#include "ctx_internal.h"

using namespace hz;
using namespace hzl;

struct hz_ctx {
    // optional per-kernel timing (HIP events on the launch stream)
    bool profiling = false;
    struct Prof { std::string name; uint64_t bytes; uint64_t units; hipEvent_t e0, e1; float ms; };
    std::vector<Prof> prof;
    size_t prof_used = 0;
    Layout lo;
    int device = 0;
    DevBuf wit, sc_tx, sc_fee, err, msg, chain, stage;
    DevBuf ed_side;   // the small-launch signature ladder's parked numerators (eddsa_side_bytes), allocated by the first launch that needs it
    // per-instance failure report (hz_witness_failures): every instance's lowest failing key, kept by report_fail beside the
    // launch-wide minimum, and -- allocated by the first call that finds failures -- the operands of each
    DevBuf inst_min, inst_rec;
    bool inst_min_dirty = false;          // a failure was seen (or no check told otherwise) since inst_min was last reset
    unsigned long long last_minkey = ~0ull;
    bool checked = false;                 // hz_witness_check has completed for the last enqueue
    std::vector<uint8_t> input_set;
    std::vector<uint8_t> host_stage;
    hipStream_t last_stream = nullptr;
    // multi-GPU intra-batch shard (RollupMain): this context evaluates transactions [sh_first, sh_first + sh_count);
    // only meaningful when `sharded`. The fee transactions and HashInputs of a sharded context run in hz_witness_enqueue_tail.
    uint32_t sh_first = 0, sh_count = 0;
    bool sharded = false;   // hz_ctx_set_shard with count >= 0 (an empty range is a legal shard: more ranks than transactions)
    bool sh_tail = true;
    // device-side input writes (hz_set_input_dev, hz_copy_instance_inputs, hz_inputs_upload) are asynchronous on the stream they were
    // given; the next enqueue waits for this event on ITS stream, whichever that is
    hipEvent_t ev_inputs = nullptr;
    bool inputs_pending = false;
    // bulk upload (hz_inputs_upload): one packed buffer per instance -> device staging -> k_unpack_inputs into the witness layout
    DevBuf upl, upl_desc, upl_bad;
    uint64_t upl_bytes = 0;                 // packed bytes of one instance
    std::vector<uint64_t> upl_off;          // per input: byte offset inside the packed buffer
    uint32_t upl_blocks = 0;
    bool upl_used = false;                  // hz_witness_check also reads the range-check flag of the unpack kernel
    // hz_inputs_stage: the H2D half alone, ahead of time (while the previous step still computes on the old inputs); the unpack
    // kernels of the staged instances run at the head of the next enqueue
    std::vector<uint8_t> staged;
    bool any_staged = false;
    hipStream_t stage_stream = nullptr;     // the stream of the stage calls since the last enqueue (ev_staged is recorded on it)
    hipStream_t s_copy = nullptr;
    hipEvent_t ev_staged = nullptr, ev_unpacked = nullptr;
    // independent chains of one batch run concurrently: the EdDSA ladders and the fee transactions on
    // their own streams, joined by events before HashInputs (DESIGN.md "Kernel schedule")
    bool exclusive = false;   // hz_ctx_set_profiling(ctx, 2)
    bool partitioned = false; // latency-bound context: CU-masked internal streams (hz_ctx_create)
    hipEvent_t ev_user_in = nullptr, ev_user_out = nullptr;
    hipStream_t s_ed = nullptr, s_fee = nullptr, s_main = nullptr;   // s_main replaces a NULL caller stream
    hipEvent_t ev_reset = nullptr, ev_front = nullptr, ev_ed = nullptr, ev_fee = nullptr, ev_fix = nullptr;
    hipEvent_t ev_sha[9] = {};   // HashInputs: chain group g done (0..7), expansion done (8)
    hipStream_t s_fix = nullptr;   // the fixed-base half of the signature check
    hipStream_t s_sha = nullptr;   // SHA-256 expansion groups behind the chain when HashInputs runs early on the fee stream
    // Constant marks. The witness buffer is persistent: what the previous step left in it is still there when the next one starts,
    // and more than a third of what a step of the headline shape used to store does not depend on its inputs -- the S-box block of
    // Poseidon(0, 0) in the hash slots of every SMT level above a proof's leaf (HZ_POSEIDON3_ZERO_WIT, 243 signals per level and
    // chain: 36.7 of 100 GB per step), zeros in the switcher / state-machine signals of those levels. k_smt keeps two bytes per
    // (chain, unit) -- from which level up the buffer holds that content -- stores an empty level only below the mark and leaves
    // the mark of what the buffer holds after the step (smt_kernels.hip). Nothing but k_smt writes those slots; the marks are cleared
    // (0xFF: "nothing held") at creation and by hz_clear_inputs. HZ_NO_ZMARK=1: no marks, every level stored every step.
    DevBuf zm_tx, zm_fee;
    DevBuf zm_skip;                // profiling: elements not stored by [0] the transaction launch, [1] the fee launch, [2] the early-tail launch
    bool zm_reset = false;         // hz_clear_inputs: the next enqueue clears the marks first
    hz::ExportScratch exp_scratch; // export.hip's per-context buffers (ctx_internal.h)
    DevBuf smt_trace;              // experiments: HZ_SMT_TRACE=<file> -- per-wavefront start / end / placement of the transaction k_smt launch
    DevBuf pos3;                   // poseidon_quad.h's constants (C, M R, M R^2): the latency form of k_smt
    bool smt_lat = false;          // this context's chain launches take the latency form (hz_ctx_create)
    bool smt_lat_few = false;      // ... while at most two partitioned contexts are alive on the device (HZ_FLAG_LATENCY, one batch)
    hipEvent_t ev_hash4 = nullptr, ev_tail = nullptr, ev_sighash = nullptr;
    void release_masked();
    ~hz_ctx() {
        if (partitioned) {   // CU-masked streams go back to the process's pool (masked_stream_pool below)
            release_masked();
            s_ed = s_fee = s_main = s_fix = nullptr;
        }
        if (s_ed) (void)hipStreamDestroy(s_ed);
        if (s_fee) (void)hipStreamDestroy(s_fee);
        if (s_main) (void)hipStreamDestroy(s_main);
        if (s_fix) (void)hipStreamDestroy(s_fix);
        if (s_copy) (void)hipStreamDestroy(s_copy);
        if (s_sha) (void)hipStreamDestroy(s_sha);
        for (hipEvent_t e : {ev_hash4, ev_tail, ev_sighash})
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : {ev_staged, ev_unpacked})
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : {ev_reset, ev_front, ev_ed, ev_fee, ev_fix, ev_user_in, ev_user_out, ev_inputs})
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_sha)
            if (e) (void)hipEventDestroy(e);
        for (auto& p : prof) { (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1); }
    }
    unsigned long long filter_stage = ~0ull;
    bool enqueued = false;
    // symbol enumeration index: cumulative symbol counts per (section, block)
    struct BlkRef { int sec, blk; uint64_t first; uint32_t units; };
    std::vector<BlkRef> sym_index;
    uint64_t sym_total = 0;
    std::string sym_name;
};

// which kernel writes a block of the tx / fee / hash-inputs sections (for algorithmic byte counts)
static const char* block_owner(const Layout& lo, int sec, const Block& b) {
    const std::string& n = b.name;
    auto has = [&](const char* t) { return n.find(t) != std::string::npos; };
    if (sec == lo.sec_hi && lo.p.tmpl != T_WITHDRAW) return "hash_inputs";
    const bool fee = (sec == lo.sec_fee);
    if (has(".hash1Old.") || has(".hash1New.") || has("St1Hash.") || has("St2Hash.") || has("StFeePck.") || has("s1OldValue") || has("s2OldValue"))
        return fee ? "fee_hash" : "hash4";
    if (has(".topSwitcher.") || has(".checkOldInput.") || has(".areKeyEquals.") || has(".keysOk.") || (has(".processor") && n.size() > 8 && n.compare(n.size() - 8, 8, ".newRoot") == 0))
        return fee ? "fee_back" : "rtx_back";
    if (has(".processor")) return fee ? "fee_smt" : "smt";
    if (!fee && lo.p.tmpl == T_ROLLUP_MAIN && has(".hashSig.")) return "sig_hash";
    if (has(".sigVerifier.mulFix.") || has(".sigVerifier.snum2bits") || has(".sigVerifier.compConstant")) return "eddsa_fix";
    if (has(".sigVerifier.eqCheck")) return "eddsa_final";
    if (has(".getAx.") || has(".sigVerifier.")) return "eddsa";
    if (has(".s3.out") || has(".s4.out") || has(".s5.out") || n == "main.hasherInputs.L1L2TxsData") return "rtx_back";
    if (!fee && lo.p.tmpl == T_ROLLUP_MAIN && has(".feeAccumulator.")) return "fee_acc";   // (standalone RollupTx: part of its front kernel)
    return fee ? "fee_front" : "front";
}
static uint64_t owner_bytes(const hz_ctx* c, const char* owner) {
    uint64_t n = 0;
    const Layout& lo = c->lo;
    if (!strcmp(owner, "withdraw") || !strcmp(owner, "withdraw_sha")) {   // the SHA part owns the sha256 blocks and the bit decompositions
        uint64_t sha = 0;
        for (size_t si = 0; si < lo.sections.size(); si++)
            for (const Block& b : lo.sections[si].blocks)
                if (b.name.find("main.hasherInputs.") == 0) sha += (uint64_t)b.count * lo.sections[si].n_units;
        return (!strcmp(owner, "withdraw_sha") ? sha : lo.total - sha) * 32;
    }
    for (size_t si = 0; si < lo.sections.size(); si++)
        for (const Block& b : lo.sections[si].blocks)
            if (!strcmp(block_owner(lo, (int)si, b), owner)) n += (uint64_t)b.count * lo.sections[si].n_units;
    return n * 32;
}
struct ProfScope {
    hz_ctx* c;
    hipStream_t s;
    bool on;
    ProfScope(hz_ctx* c_, hipStream_t s_, const char* name, uint64_t units) : c(c_), s(s_), on(c_->profiling) {
        if (!on) return;
        if (c->prof_used == c->prof.size()) {
            hz_ctx::Prof p;
            p.name = name; p.bytes = owner_bytes(c, name); p.units = units; p.ms = 0;
            (void)hipEventCreate(&p.e0);
            (void)hipEventCreate(&p.e1);
            c->prof.push_back(p);
        }
        (void)hipEventRecord(c->prof[c->prof_used].e0, s);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(c->prof[c->prof_used].e1, s);
        c->prof_used++;
    }
};


// Every CU-masked stream owns a hardware queue, and the runtime does not hand a destroyed one's queue back soon enough: a process that
// creates and destroys HZ_FLAG_LATENCY contexts in a loop (a benchmark's sweep, a test suite) ran out of queues after a few dozen
// (HSA_STATUS_ERROR_OUT_OF_RESOURCES, the process aborts). The four streams of a partitioned context are therefore never destroyed:
// a context that ends puts them into this pool (per device and role: the mask is a function of both), the next one takes them from it.
// A stream in the pool is idle: the context synchronised with its work before it let go.
// The other half of the same hazard is NOT cured here: with ROCr's asynchronous scratch reclaim (the default of ROCm 7), a process that
// has made plain contexts and then keeps four HZ_FLAG_LATENCY contexts of the headline shape in flight -- sixteen hardware queues whose
// kernels use scratch (k_main_front 7.7 KB per lane) -- can abort in a queue callback (HSA_STATUS_ERROR_OUT_OF_RESOURCES with 247 GB of
// HBM free; tools/experiments/masked_stream_churn.py reproduces it in seconds). HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0 cures it
// (tools/experiments/scratch_env_probe.sh: the only one of six runtime knobs that does) and costs 10-50 % of every step (the headline
// 36.2 -> 39.9 ms, 16 batches x 2 contexts 20.9 -> 45.9 ms): not a default. A fresh process with up to four such contexts has not
// failed; bench.py runs those sweep points in child processes. The cure proper is kernels without scratch.
namespace {
#ifndef HZ_MAX_PARTITIONED
#define HZ_MAX_PARTITIONED 4   // (2 until round 6: with k_main_front at 7.7 KB of scratch per lane four flagged contexts in flight aborted in the runtime)
#endif
struct MaskedPool {
    std::mutex mu;
    std::vector<hipStream_t> idle[16][4];
    int alive[16] = {};   // partitioned contexts per device
};
MaskedPool& masked_pool() { static MaskedPool* p = new MaskedPool(); return *p; }   // (never destroyed: streams outlive static teardown order)
}  // namespace
void hz_ctx::release_masked() {
    if (device < 0 || device >= 16) return;
    hipStream_t st[4] = {s_ed, s_fix, s_fee, s_main};
    MaskedPool& P = masked_pool();
    {
        std::lock_guard<std::mutex> g(P.mu);
        if (P.alive[device] > 0) P.alive[device]--;
    }
    for (int k = 0; k < 4; k++) {
        if (!st[k]) continue;
        (void)hipStreamSynchronize(st[k]);
        std::lock_guard<std::mutex> g(P.mu);
        P.idle[device][k].push_back(st[k]);
    }
}

static uint8_t* sec_ptr(hz_ctx* c, int sec) { return (uint8_t*)c->wit.p + c->lo.sections[sec].base * 32; }

static uint32_t block_units(const Layout&, const Section& s, const Block& b) {
    return (b.max_units >= 0) ? (uint32_t)b.max_units : s.upi;   // symbols describe one instance
}

extern "C" hz_status hz_ctx_create(const hz_params* p, hz_ctx** out) {
    if (!p || !out) return set_err(HZ_ERR_ARG, "hz_ctx_create: null argument");
    if (p->template_id < 0 || p->template_id >= T_COUNT) return set_err(HZ_ERR_ARG, "hz_ctx_create: unknown template %d", p->template_id);
    Params lp;
    lp.tmpl = p->template_id; lp.nTx = p->nTx; lp.L = p->nLevels; lp.maxL1 = p->maxL1Tx; lp.F = p->maxFeeTx;
    lp.n_inst = p->n_instances > 0 ? p->n_instances : 1;
    const bool needs_L = (lp.tmpl != T_HASH_STATE && lp.tmpl < T_DECODE_FLOAT) || lp.tmpl == T_SMT_PROCESSOR || lp.tmpl == T_SMT_VERIFIER;
    if (lp.tmpl == T_FEE_ACCUMULATOR && lp.F < 1) return set_err(HZ_ERR_ARG, "FeeAccumulator needs maxFeeTx >= 1");
    if (needs_L && (lp.L < 2 || lp.L > 48)) return set_err(HZ_ERR_ARG, "nLevels must be in [2,48]");
    if ((lp.tmpl == T_ROLLUP_MAIN || lp.tmpl == T_HASH_INPUTS) && (lp.nTx < 1 || lp.F < 1 || lp.maxL1 < 0))
        return set_err(HZ_ERR_ARG, "RollupMain/HashInputs need nTx >= 1, maxFeeTx >= 1");
    if (lp.tmpl == T_ROLLUP_TX && lp.F < 1) return set_err(HZ_ERR_ARG, "RollupTx needs maxFeeTx >= 1");
    if (lp.tmpl == T_ROLLUP_MAIN && lp.maxL1 > lp.nTx) return set_err(HZ_ERR_ARG, "RollupMain: maxL1Tx (%d) > nTx (%d) cannot be instantiated", lp.maxL1, lp.nTx);
    {
        // the failure key holds instance and unit in 24 bits each; unit counts are 32-bit
        const uint64_t upi = (lp.tmpl == T_ROLLUP_MAIN || lp.tmpl == T_HASH_INPUTS) ? (uint64_t)std::max(lp.nTx, lp.F) : 1;
        if ((uint64_t)lp.n_inst >= (1u << 24) || upi >= (1u << 24) || (uint64_t)lp.n_inst * upi >= (1ull << 31))
            return set_err(HZ_ERR_ARG, "hz_ctx_create: n_instances (%d) x units per instance (%llu) out of range", lp.n_inst, (unsigned long long)upi);
    }
    if (hz_device_count() <= 0) return set_err(HZ_ERR_NODEVICE, "no usable gfx950 device");
    if (p->device < 0 || p->device >= hz_device_count()) return set_err(HZ_ERR_ARG, "bad device ordinal %d", p->device);
    hz_ctx* c = new hz_ctx();
    build_layout(lp, c->lo);
    c->device = p->device;
    hipError_t e = hipSetDevice(c->device);
    const Layout& lo = c->lo;
    if (e == hipSuccess) e = c->wit.alloc(lo.total * 32);
    if (e == hipSuccess) e = hipMemset(c->wit.p, 0, lo.total * 32);
    if (e == hipSuccess) e = c->err.alloc(sizeof(ErrBuf));
    {
        // k_smt's latency form (a quad of lanes per chain: 0.63 x the time of a dependent level hash for ~2.5 x its instructions, one
        // wavefront per SIMD) -- where the chain IS the step and the step is small: the standalone SMTProcessor and FeeTx mains of at
        // most HZ_SMT_LAT_MAX units. Not RollupTx / RollupMain: their single batch is bound by the signature chain (9.5 ms against
        // 8.0-8.3 for the state-tree chain with this form, 9.5 without), and with several contexts in flight its 512 wavefronts at one
        // per SIMD hold a whole CU partition -- four HZ_FLAG_LATENCY contexts of one batch each: 432 k tx/s with it, 560 k without; a
        // small launch beside big ones (the fee chain of a 32-batch step) cost the headline 5 % (profiles/r05_smt_latency_form.txt).
        // HZ_SMT_LATENCY_FORM=1 forces it for every context whose step is small enough (tests, experiments), =0 forbids it.
        uint32_t step_units = 0;
        for (const auto& sec : lo.sections) step_units = std::max(step_units, (uint32_t)sec.n_units);
        const char* force = getenv("HZ_SMT_LATENCY_FORM");
        // ... and HZ_FLAG_SOLO contexts: the caller says nothing else runs on the device while this context's step does, and since the
        // doubling chain left the signature prologue the state-tree chain is the longer one of a single batch (9.6 against 8.2 ms)
        const bool chain_is_step = lo.p.tmpl == T_SMT_PROCESSOR || lo.p.tmpl == T_FEE_TX || (p->flags & HZ_FLAG_SOLO) != 0;
        c->smt_lat = step_units <= HZ_SMT_LAT_MAX && (force ? force[0] == '1' : chain_is_step);
        // ... and HZ_FLAG_LATENCY contexts of one batch WHILE AT MOST TWO of them are alive on the device (decided per launch,
        // enqueue_smt_chain): one alone 8.7 -> 7.7 ms, two in flight 427-436 k -> 448 k tx/s, four in flight 640 k -> 540 k
        // (profiles/r06_flagged_latency_form.txt) -- the form's 512 wavefronts at one per SIMD fill the main stream's partition once,
        // two contexts' launches rarely coincide, four contexts' do.
        c->smt_lat_few = !c->smt_lat && !force && step_units <= HZ_SMT_LAT_MAX && lo.p.tmpl == T_ROLLUP_MAIN &&
                         ((p->flags & HZ_FLAG_LATENCY) != 0 || getenv("HZ_FORCE_LATENCY_SCHEDULING") != nullptr);   // (the same condition as `want` below)
        if (e == hipSuccess && (c->smt_lat || c->smt_lat_few)) e = c->pos3.alloc(pos3_dense_bytes());
        if (e == hipSuccess && c->pos3.p) e = upload_pos3_dense((Fr*)c->pos3.p);
    }
    if (e == hipSuccess) e = c->inst_min.alloc((size_t)lo.n_inst * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(c->inst_min.p, 0xFF, c->inst_min.bytes);
    if (e == hipSuccess) e = hipMemset(c->err.p, 0, offsetof(ErrBuf, rec));
    if (e == hipSuccess) e = hipMemcpy((uint8_t*)c->err.p + offsetof(ErrBuf, inst_min), &c->inst_min.p, sizeof(void*), hipMemcpyHostToDevice);
    if (e == hipSuccess && lo.sec_tx >= 0) e = c->sc_tx.alloc((size_t)SC_COUNT * lo.sections[lo.sec_tx].n_units * sizeof(Fr));
    if (e == hipSuccess && lo.sec_fee >= 0) e = c->sc_fee.alloc((size_t)SC_COUNT * lo.sections[lo.sec_fee].n_units * sizeof(Fr));
    if (!getenv("HZ_NO_ZMARK")) {   // constant marks: two bytes per (chain, unit) of every section k_smt walks
        if (e == hipSuccess && lo.sec_tx >= 0) e = c->zm_tx.alloc((size_t)4 * lo.sections[lo.sec_tx].n_units * sizeof(uint16_t));
        if (e == hipSuccess && lo.sec_fee >= 0) e = c->zm_fee.alloc((size_t)2 * lo.sections[lo.sec_fee].n_units * sizeof(uint16_t));
        if (e == hipSuccess && c->zm_tx.p) e = hipMemset(c->zm_tx.p, 0xFF, c->zm_tx.bytes);
        if (e == hipSuccess && c->zm_fee.p) e = hipMemset(c->zm_fee.p, 0xFF, c->zm_fee.bytes);
        if (e == hipSuccess) e = c->zm_skip.alloc(3 * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMemset(c->zm_skip.p, 0, c->zm_skip.bytes);
    }
    if (e == hipSuccess && lo.sec_hi >= 0) {
        e = c->msg.alloc((size_t)lo.hi.sha.nblocks * 64 * lo.n_inst);
        if (e == hipSuccess) e = c->chain.alloc((size_t)(lo.hi.sha.nblocks + 1) * 32 * lo.n_inst);
    }
    {
        // A context that is latency bound (one to four batches: a few dozen wavefronts per kernel) gives each of its concurrent
        // chains its own compute units through CU-masked streams: kernels that share a CU share its instruction cache and issue
        // slots, and the 213 KB k_smt, the signature kernels and the fee chain evict each other (single batch: 36.5 -> 25 ms).
        // Throughput-sized contexts keep unmasked streams: the integer pipe is their limit and every CU should take any work.
        hipDeviceProp_t prop;
        const uint64_t units = (uint64_t)lo.n_inst * (lo.sec_tx >= 0 ? lo.sections[lo.sec_tx].upi : 0);
        // Opt-in (HZ_FLAG_LATENCY). Measured, tx/s with one context: 1 batch 56 k -> 81 k (36.5 -> 25 ms), 4: 228 k -> 334 k,
        // 8: 390 k -> 479 k, 16: 448 k -> 575 k; with two contexts in flight the partition loses (32 x 2: 1007 k -> 662 k).
        // Not automatic: every CU-masked stream owns a hardware queue, and a process that started while another one had
        // created and destroyed ~130 of them blocked in its first launch (tests with HZ_FORCE_LATENCY_SCHEDULING).
        (void)units;
        const bool want = (p->flags & HZ_FLAG_LATENCY) != 0 || getenv("HZ_FORCE_LATENCY_SCHEDULING") != nullptr;
        c->partitioned = lo.p.tmpl == T_ROLLUP_MAIN && want && hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount >= 64;
        // At most FOUR partitioned contexts alive per device and process by default (HZ_MAX_PARTITIONED=<n> changes it): each holds four
        // hardware queues, and the scratch memory of a queue is sized by the hungriest kernel it has run. Round 5 capped this at two: with
        // k_main_front at 7.7 KB of scratch per lane, four flagged contexts in flight (sixteen queues) aborted in the runtime in a process
        // that had made plain contexts before. Round 6 took that kernel to 480 bytes (the largest of a flagged context's kernels is now
        // k_eddsa_ladder<1> at 2.8 KB) and the reproducer (tools/experiments/masked_stream_churn.py 3 1: plain, 2, 4, 3 flagged contexts
        // of the headline shape, three rounds) runs clean; tests/test_witness_gpu.py keeps it as a test, in a process of its own.
        // A context over the limit gets plain streams: the same witness, the default schedule.
        if (c->partitioned && c->device >= 0 && c->device < 16) {
            static const int cap = getenv("HZ_MAX_PARTITIONED") ? atoi(getenv("HZ_MAX_PARTITIONED")) : HZ_MAX_PARTITIONED;
            MaskedPool& P = masked_pool();
            std::lock_guard<std::mutex> g(P.mu);
            if (P.alive[c->device] >= cap) c->partitioned = false;
            else P.alive[c->device]++;
        }
        const int ncu = c->partitioned ? prop.multiProcessorCount : 0;
        // The variable-base ladder is the longest dependent chain of a step and runs on few wavefronts: its stream gets the highest
        // priority, so its workgroups are dispatched ahead of the wide kernels' (a 32-batch step alone: 56.8 -> 47.9 ms; two contexts
        // in flight: unchanged, 44 ms per step).
        int prio_least = 0, prio_greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        auto make_stream = [&](hipStream_t* st, int lo_cu, int hi_cu) {
            if (!c->partitioned) {
                static const char* dq = getenv("HZ_DEDICATED_QUEUES");   // experiment: a hardware queue per stream (a CU mask of every CU), no partition
                if (dq && dq[0] == '1' && hipGetDeviceProperties(&prop, c->device) == hipSuccess) {
                    if (dq[1] == 'e' && st != &c->s_ed) return hipStreamCreateWithFlags(st, hipStreamNonBlocking);   // "1e": the ladder stream only
                    std::vector<uint32_t> all((prop.multiProcessorCount + 31) / 32, 0u);
                    for (int b = 0; b < prop.multiProcessorCount; b++) all[b >> 5] |= 1u << (b & 31);
                    return hipExtStreamCreateWithCUMask(st, (uint32_t)all.size(), all.data());
                }
                if (st == &c->s_ed && prio_greatest != prio_least) return hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio_greatest);
                return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
            }
            if (c->device >= 0 && c->device < 16) {   // an idle one of the same role, if the process has made one before
                const int role = st == &c->s_ed ? 0 : st == &c->s_fix ? 1 : st == &c->s_fee ? 2 : 3;
                MaskedPool& P = masked_pool();
                std::lock_guard<std::mutex> g(P.mu);
                if (!P.idle[c->device][role].empty()) {
                    *st = P.idle[c->device][role].back();
                    P.idle[c->device][role].pop_back();
                    return hipSuccess;
                }
            }
            std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
            for (int b = lo_cu; b < hi_cu; b++) mask[b >> 5] |= 1u << (b & 31);
            return hipExtStreamCreateWithCUMask(st, (uint32_t)mask.size(), mask.data());
        };
        if (e == hipSuccess) e = make_stream(&c->s_ed, 0, ncu / 4);                  // variable-base ladder
        if (e == hipSuccess) e = make_stream(&c->s_fix, ncu / 4, ncu * 3 / 8);       // fixed-base half
        if (e == hipSuccess) e = make_stream(&c->s_fee, ncu * 3 / 8, ncu / 2);       // fee-transaction chain
        if (e == hipSuccess) e = make_stream(&c->s_main, ncu / 2, ncu);              // front, hash-state, SMT chains, HashInputs
        if (e == hipSuccess && lo.p.tmpl == T_ROLLUP_MAIN && !c->partitioned) e = make_stream(&c->s_sha, 0, 0);
        for (hipEvent_t* ev : {&c->ev_hash4, &c->ev_tail, &c->ev_sighash})
            if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
    }
    for (hipEvent_t* ev : {&c->ev_reset, &c->ev_front, &c->ev_ed, &c->ev_fee, &c->ev_fix, &c->ev_user_in, &c->ev_user_out, &c->ev_inputs})
        if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
    if (lo.sec_hi >= 0)
        for (hipEvent_t& ev : c->ev_sha)
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) {
        delete c;
        return set_err(HZ_ERR_HIP, "hz_ctx_create: %s (witness buffer %.1f MiB)", hipGetErrorString(e), lo.total * 32.0 / 1048576.0);
    }
    c->input_set.assign(lo.inputs.size(), 0);
    for (size_t si = 0; si < lo.sections.size(); si++)
        for (size_t bi = 0; bi < lo.sections[si].blocks.size(); bi++) {
            const Block& b = lo.sections[si].blocks[bi];
            const uint32_t units = block_units(lo, lo.sections[si], b);
            c->sym_index.push_back({(int)si, (int)bi, c->sym_total, units});
            c->sym_total += (uint64_t)units * b.count;
        }
    *out = c;
    return HZ_OK;
}

extern "C" void hz_ctx_destroy(hz_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->smt_trace.p && getenv("HZ_SMT_TRACE")) {   // the LAST transaction launch's trace, appended to the file (one record per context)
        (void)hipDeviceSynchronize();
        std::vector<uint8_t> h(c->smt_trace.bytes);
        if (hipMemcpy(h.data(), c->smt_trace.p, h.size(), hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE* f = fopen(getenv("HZ_SMT_TRACE"), "ab")) { const uint64_t n = h.size(); fwrite(&n, 8, 1, f); fwrite(h.data(), 1, h.size(), f); fclose(f); }
        }
    }
    delete c;
}
extern "C" uint64_t hz_witness_len(const hz_ctx* c) { return c ? c->lo.per_instance : 0; }
extern "C" int32_t hz_ctx_ntx(const hz_ctx* c) { return (c && c->lo.p.tmpl == T_ROLLUP_MAIN) ? (int32_t)c->lo.p.nTx : 0; }
// Device memory a context of this layout holds once every buffer that is allocated on first use exists (capacity planning: how many
// batches fit the 288 GB of a device). The same arithmetic as hz_ctx_create and the lazy allocations below.
static uint64_t layout_device_bytes(const Layout& lo) {
    uint64_t n = lo.total * 32 + sizeof(ErrBuf) + (uint64_t)lo.n_inst * (sizeof(unsigned long long) + sizeof(ErrRec));
    if (lo.sec_tx >= 0) n += (uint64_t)(SC_COUNT * sizeof(Fr) + 4 * sizeof(uint16_t)) * lo.sections[lo.sec_tx].n_units;
    if (lo.p.tmpl == T_ROLLUP_MAIN || lo.p.tmpl == T_ROLLUP_TX) n += eddsa_side_bytes(lo.sections[lo.sec_tx].n_units);
    if (lo.sec_fee >= 0) n += (uint64_t)(SC_COUNT * sizeof(Fr) + 2 * sizeof(uint16_t)) * lo.sections[lo.sec_fee].n_units;
    if (lo.sec_hi >= 0) n += ((uint64_t)lo.hi.sha.nblocks * 64 + ((uint64_t)lo.hi.sha.nblocks + 1) * 32) * lo.n_inst;
    uint64_t packed = 0;   // the staging slots of the bulk-upload path (hz_inputs_upload / hz_inputs_stage), one per instance
    for (const InputDesc& d : lo.inputs) packed = ((packed + 31) & ~31ull) + (uint64_t)d.inner * d.outer * d.ebytes;
    return n + ((packed + 31) & ~31ull) * lo.n_inst;
}
extern "C" uint64_t hz_ctx_device_bytes(const hz_ctx* c) { return c ? layout_device_bytes(c->lo) : 0; }
extern "C" uint64_t hz_template_device_bytes(const hz_params* p) {
    if (!p || p->template_id < 0 || p->template_id >= T_COUNT) return 0;
    Params lp;
    lp.tmpl = p->template_id; lp.nTx = p->nTx; lp.L = p->nLevels; lp.maxL1 = p->maxL1Tx; lp.F = p->maxFeeTx;
    lp.n_inst = p->n_instances > 0 ? p->n_instances : 1;
    Layout lo;
    build_layout(lp, lo);
    return layout_device_bytes(lo);
}
extern "C" uint64_t hz_constraint_estimate(const hz_ctx* c) { return c ? constraint_estimate(c->lo.p) : 0; }
extern "C" const void* hz_witness_dev_ptr(const hz_ctx* c) { return c ? c->wit.p : nullptr; }
extern "C" int32_t hz_input_count(const hz_ctx* c) { return c ? (int32_t)c->lo.inputs.size() : 0; }
extern "C" const char* hz_input_name(const hz_ctx* c, int32_t i, uint64_t* flat_len) {
    if (!c || i < 0 || (size_t)i >= c->lo.inputs.size()) return nullptr;
    const InputDesc& d = c->lo.inputs[i];
    if (flat_len) *flat_len = (uint64_t)d.inner * d.outer;
    return d.name.c_str();
}
extern "C" void hz_clear_inputs(hz_ctx* c) {
    if (!c) return;
    std::fill(c->input_set.begin(), c->input_set.end(), 0);
    c->zm_reset = true;   // a caller that starts over gets a step that stores everything
}
extern "C" const char* hz_constraint_name(int32_t id) { return constraint_name(id); }

// transpose [outer][inner] -> [inner][outer] on the device (hz_set_input_dev)
__global__ void k_transpose32(const uint4* __restrict__ src, uint4* __restrict__ dst, uint32_t outer, uint32_t inner, uint32_t dst_pitch_elems) {
    const size_t n = (size_t)outer * inner;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const uint32_t k = (uint32_t)(t / outer), u = (uint32_t)(t % outer);   // consecutive lanes -> consecutive units (coalesced stores)
        const size_t s = ((size_t)u * inner + k) * 2, d = ((size_t)k * dst_pitch_elems + u) * 2;
        dst[d] = src[s];
        dst[d + 1] = src[s + 1];
    }
}

static hz_status set_input_common(hz_ctx* c, int32_t instance, const char* name, const uint8_t* host, const void* dev, size_t count, hipStream_t stream) {
    if (!c || !name || (!host && !dev)) return set_err(HZ_ERR_ARG, "hz_set_input: null argument");
    const Layout& lo = c->lo;
    const InputDesc* d = lo.find_input(name);
    if (!d) return set_err(HZ_ERR_INPUT, "Signal not found: %s", name);
    const Section& s = lo.sections[d->section];
    HZ_HIP(hipSetDevice(c->device));
    if (host)
        for (size_t i = 0; i < count; i++)
            if (!canon_lt_p(host + 32 * i)) return set_err(HZ_ERR_INPUT, "input %s[%zu] is not a canonical field element (>= r)", name, i);
    uint8_t* dst0 = sec_ptr(c, d->section) + (size_t)d->off * s.n_units * 32;   // element (off, unit 0)
    const size_t pitch = (size_t)s.n_units * 32;
    const uint32_t B = lo.n_inst, outer = d->outer;
    const size_t per = (size_t)d->inner * outer;
    uint32_t b0, b1;
    if (instance >= 0) {
        if ((uint32_t)instance >= B || count != per)
            return set_err(HZ_ERR_INPUT, "input %s: expected %zu values for instance %d, got %zu", name, per, instance, count);
        b0 = (uint32_t)instance; b1 = b0 + 1;
    } else {
        if (count != per * B)
            return set_err(HZ_ERR_INPUT, "input %s: expected %zu values for all %u instances, got %zu", name, per * B, B, count);
        b0 = 0; b1 = B;
    }
    if (host) {
        // host transpose [instance][outer][inner] -> [inner][instance][outer], then one 2-D copy per instance
        // (a single copy when the signal covers every unit of the section)
        const uint32_t nb = b1 - b0;
        c->host_stage.resize(count * 32);
        uint8_t* st = c->host_stage.data();
        for (uint32_t b = 0; b < nb; b++)
            for (uint32_t u = 0; u < outer; u++)
                for (uint32_t k = 0; k < d->inner; k++)
                    memcpy(st + (((size_t)k * nb + b) * outer + u) * 32, host + (((size_t)b * outer + u) * d->inner + k) * 32, 32);
        if (outer == s.upi) {
            HZ_HIP(hipMemcpy2D(dst0 + (size_t)b0 * s.upi * 32, pitch, st, (size_t)nb * outer * 32, (size_t)nb * outer * 32, d->inner, hipMemcpyHostToDevice));
        } else {
            for (uint32_t b = 0; b < nb; b++)
                HZ_HIP(hipMemcpy2D(dst0 + (size_t)(b0 + b) * s.upi * 32, pitch, st + (size_t)b * outer * 32, (size_t)nb * outer * 32, (size_t)outer * 32, d->inner,
                                   hipMemcpyHostToDevice));
        }
    } else {
        if (!stream) stream = c->s_main;   // never the legacy default stream (hermez_witness.h "streams")
        for (uint32_t b = b0; b < b1; b++) {
            const uint8_t* src = (const uint8_t*)dev + (size_t)(b - b0) * per * 32;
            uint8_t* dst = dst0 + (size_t)b * s.upi * 32;
            if (d->inner == 1 || outer == 1) {
                HZ_HIP(hipMemcpy2DAsync(dst, pitch, src, (size_t)outer * 32, (size_t)outer * 32, d->inner, hipMemcpyDeviceToDevice, stream));
            } else {
                const size_t n = (size_t)outer * d->inner;
                const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
                hipLaunchKernelGGL(k_transpose32, dim3(blocks), dim3(256), 0, stream, (const uint4*)src, (uint4*)dst, outer, d->inner, s.n_units);
                HZ_HIP(hipGetLastError());
            }
        }
        HZ_HIP(hipEventRecord(c->ev_inputs, stream));
        c->inputs_pending = true;
    }
    c->input_set[d - &lo.inputs[0]] = 1;
    return HZ_OK;
}

extern "C" hz_status hz_set_input(hz_ctx* c, int32_t instance, const char* name, const uint8_t* vals, size_t count) {
    return set_input_common(c, instance, name, vals, nullptr, count, nullptr);
}
extern "C" hz_status hz_set_input_dev(hz_ctx* c, int32_t instance, const char* name, const void* dvals, size_t count, void* stream) {
    return set_input_common(c, instance, name, nullptr, dvals, count, (hipStream_t)stream);
}

// Copy every input signal of instance `src` onto instance `dst` (device to device). Serving code that
// proves the same batch shape repeatedly, and the benchmark, fill instance 0 once and replicate.
extern "C" hz_status hz_copy_instance_inputs(hz_ctx* c, int32_t src, int32_t dst, void* stream) {
    if (!c) return set_err(HZ_ERR_ARG, "hz_copy_instance_inputs: null context");
    const Layout& lo = c->lo;
    if (src < 0 || dst < 0 || (uint32_t)src >= lo.n_inst || (uint32_t)dst >= lo.n_inst)
        return set_err(HZ_ERR_ARG, "hz_copy_instance_inputs: instance out of range (n_instances = %u)", lo.n_inst);
    if (src == dst) return HZ_OK;
    HZ_HIP(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->s_main;
    for (const InputDesc& d : lo.inputs) {
        const Section& s = lo.sections[d.section];
        uint8_t* base = sec_ptr(c, d.section) + (size_t)d.off * s.n_units * 32;
        const size_t pitch = (size_t)s.n_units * 32;
        HZ_HIP(hipMemcpy2DAsync(base + (size_t)dst * s.upi * 32, pitch, base + (size_t)src * s.upi * 32, pitch, (size_t)d.outer * 32, d.inner,
                                hipMemcpyDeviceToDevice, st));
    }
    HZ_HIP(hipEventRecord(c->ev_inputs, st));
    c->inputs_pending = true;
    return HZ_OK;
}

// ---- bulk input path ---------------------------------------------------------------------------------------
// The per-signal hz_set_input walks host memory three times (range check, transpose, pageable copy). A serving process that feeds
// batch after batch hands over ONE packed buffer per instance instead -- every input signal in hz_input_name order, each as
// [outer][inner] little-endian elements of hz_input_packed_width bytes (what hz_set_input takes, concatenated; bit signals one
// byte each) -- ideally in pinned memory (hz_host_alloc): one asynchronous H2D copy into a device staging slot, then one kernel
// transposes every signal into the witness layout and range-checks the elements (< r) on the way.
struct UnpackDesc { uint64_t src_off, dst_elem0; uint32_t inner, outer, ebytes, n_units, upi, first_block, index, pad; };
struct UnpackBad { unsigned long long key; };   // ~0 = none; else (input index << 40) | element
__global__ __launch_bounds__(256) void k_unpack_inputs(const UnpackDesc* __restrict__ desc, uint32_t n_desc, const uint8_t* __restrict__ src0,
                                                       uint8_t* __restrict__ wit, uint32_t inst0, uint64_t slot_bytes, UnpackBad* bad) {
    // blockIdx.y: consecutive instances from consecutive staging slots in one launch (all staged instances of a step)
    const uint32_t inst = inst0 + blockIdx.y;
    const uint8_t* src = src0 + (uint64_t)blockIdx.y * slot_bytes;
    uint32_t lo = 0, hi = n_desc - 1;   // the input this block belongs to
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (desc[mid].first_block <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const UnpackDesc d = desc[lo];
    const uint64_t t = (uint64_t)(blockIdx.x - d.first_block) * 256 + threadIdx.x;
    if (t >= (uint64_t)d.inner * d.outer) return;
    const uint32_t k = (uint32_t)(t / d.outer), u = (uint32_t)(t % d.outer);   // consecutive lanes -> consecutive units: coalesced stores
    const uint8_t* sp = src + d.src_off + ((uint64_t)u * d.inner + k) * d.ebytes;
    uint4 a, b;
    if (d.ebytes == 32) {
        a = reinterpret_cast<const uint4*>(sp)[0];
        b = reinterpret_cast<const uint4*>(sp)[1];
        // < r ? (compare from the top limb)
        const uint32_t P[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        bool lt = false, decided = false;
        for (int i = 7; i >= 0; i--)
            if (!decided && v[i] != P[i]) { lt = v[i] < P[i]; decided = true; }
        if (!lt) atomicMin(&bad->key, ((unsigned long long)d.index << 40) | t);
    } else {
        a = make_uint4(sp[0], 0u, 0u, 0u);
        b = make_uint4(0u, 0u, 0u, 0u);
    }
    uint4* dp = reinterpret_cast<uint4*>(wit + (d.dst_elem0 + (uint64_t)k * d.n_units + (uint64_t)inst * d.upi + u) * 32);
    dp[0] = a;
    dp[1] = b;
}

static hz_status upload_prepare(hz_ctx* c) {
    if (c->upl_desc.p) return HZ_OK;
    const Layout& lo = c->lo;
    std::vector<UnpackDesc> ds;
    uint64_t off = 0;
    uint32_t blocks = 0;
    c->upl_off.clear();
    for (size_t i = 0; i < lo.inputs.size(); i++) {
        const InputDesc& d = lo.inputs[i];
        const Section& s = lo.sections[d.section];
        off = (off + 31) & ~31ull;   // every signal starts 32-byte aligned (vector loads)
        UnpackDesc u;
        memset(&u, 0, sizeof u);
        u.src_off = off; u.dst_elem0 = s.base + (uint64_t)d.off * s.n_units; u.inner = d.inner; u.outer = d.outer; u.ebytes = d.ebytes;
        u.n_units = s.n_units; u.upi = s.upi; u.first_block = blocks; u.index = (uint32_t)i;
        ds.push_back(u);
        c->upl_off.push_back(off);
        off += (uint64_t)d.inner * d.outer * d.ebytes;
        blocks += (uint32_t)(((uint64_t)d.inner * d.outer + 255) / 256);
    }
    c->upl_bytes = (off + 31) & ~31ull;
    c->upl_blocks = blocks;
    HZ_HIP(c->upl_desc.alloc(ds.size() * sizeof(UnpackDesc)));
    HZ_HIP(hipMemcpy(c->upl_desc.p, ds.data(), ds.size() * sizeof(UnpackDesc), hipMemcpyHostToDevice));
    HZ_HIP(c->upl_bad.alloc(sizeof(UnpackBad)));
    HZ_HIP(hipMemset(c->upl_bad.p, 0xFF, sizeof(UnpackBad)));
    return HZ_OK;
}
extern "C" uint64_t hz_inputs_packed_bytes(const hz_ctx* cc) {
    hz_ctx* c = const_cast<hz_ctx*>(cc);
    if (!c || hipSetDevice(c->device) != hipSuccess || upload_prepare(c) != HZ_OK) return 0;
    return c->upl_bytes;
}
extern "C" int32_t hz_input_packed_width(const hz_ctx* c, int32_t i) {
    return (!c || i < 0 || (size_t)i >= c->lo.inputs.size()) ? 0 : (int32_t)c->lo.inputs[i].ebytes;
}
extern "C" uint64_t hz_input_packed_offset(const hz_ctx* cc, int32_t i) {
    hz_ctx* c = const_cast<hz_ctx*>(cc);
    if (!c || i < 0 || (size_t)i >= c->lo.inputs.size() || hipSetDevice(c->device) != hipSuccess || upload_prepare(c) != HZ_OK) return ~0ull;
    return c->upl_off[i];
}
extern "C" void* hz_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)set_err(HZ_ERR_HIP, "hz_host_alloc: %zu bytes of pinned memory", bytes); return nullptr; }
    return p;
}
extern "C" void hz_host_free(void* p) { if (p) (void)hipHostFree(p); }
static hz_status upload_common(hz_ctx* c, const char* who, int32_t instance, const void* packed, size_t bytes, uint8_t** slot) {
    if (!c || !packed) return set_err(HZ_ERR_ARG, "%s: null argument", who);
    HZ_HIP(hipSetDevice(c->device));
    const hz_status st = upload_prepare(c);
    if (st != HZ_OK) return st;
    const Layout& lo = c->lo;
    if (instance < 0 || (uint32_t)instance >= lo.n_inst) return set_err(HZ_ERR_ARG, "%s: instance %d out of range (n_instances = %u)", who, instance, lo.n_inst);
    if (bytes != c->upl_bytes) return set_err(HZ_ERR_INPUT, "%s: expected %llu packed bytes, got %zu", who, (unsigned long long)c->upl_bytes, bytes);
    if (!c->upl.p) HZ_HIP(c->upl.alloc((size_t)c->upl_bytes * lo.n_inst));
    *slot = (uint8_t*)c->upl.p + (size_t)instance * c->upl_bytes;
    return HZ_OK;
}
static hz_status launch_unpack(hz_ctx* c, uint32_t instance, hipStream_t s, uint32_t count = 1) {
    const uint8_t* slot = (const uint8_t*)c->upl.p + (size_t)instance * c->upl_bytes;
    hipLaunchKernelGGL(k_unpack_inputs, dim3(c->upl_blocks, count), dim3(256), 0, s, (const UnpackDesc*)c->upl_desc.p, (uint32_t)c->lo.inputs.size(), slot,
                       (uint8_t*)c->wit.p, instance, (uint64_t)c->upl_bytes, (UnpackBad*)c->upl_bad.p);
    HZ_HIP(hipGetLastError());
    c->upl_used = true;
    std::fill(c->input_set.begin(), c->input_set.end(), 1);   // like hz_set_input: "set" is tracked per signal, not per instance
    return HZ_OK;
}
extern "C" hz_status hz_inputs_upload(hz_ctx* c, int32_t instance, const void* packed, size_t bytes, void* stream) {
    uint8_t* slot = nullptr;
    hz_status st = upload_common(c, "hz_inputs_upload", instance, packed, bytes, &slot);
    if (st != HZ_OK) return st;
    hipStream_t s = stream ? (hipStream_t)stream : c->s_main;
    if (c->ev_unpacked) HZ_HIP(hipStreamWaitEvent(s, c->ev_unpacked, 0));   // a staged copy of this slot may still be waiting for its unpack
    HZ_HIP(hipMemcpyAsync(slot, packed, bytes, hipMemcpyDefault, s));   // (`packed` may be device memory: inputs kept resident in packed form)
    st = launch_unpack(c, (uint32_t)instance, s);
    if (st != HZ_OK) return st;
    HZ_HIP(hipEventRecord(c->ev_inputs, s));
    c->inputs_pending = true;
    return HZ_OK;
}
// The copy alone. Typical use: right after hz_witness_enqueue of step N, stage the inputs of step N + 1 -- the PCIe transfer
// runs beside step N's kernels (which still read the old inputs from the witness buffer); the next hz_witness_enqueue scatters
// every staged instance into the witness layout first.
extern "C" hz_status hz_inputs_stage(hz_ctx* c, int32_t instance, const void* packed, size_t bytes, void* stream) {
    uint8_t* slot = nullptr;
    const hz_status st = upload_common(c, "hz_inputs_stage", instance, packed, bytes, &slot);
    if (st != HZ_OK) return st;
    if (!c->ev_staged) {
        // the copy stream. A partitioned context (HZ_FLAG_LATENCY) stages on its MAIN stream instead: it already holds four hardware
        // queues, four such contexts hold sixteen -- all the runtime is given (GPU_MAX_HW_QUEUES) -- and a fifth queue each made the
        // scheduler time-slice them (four contexts in flight: 94 ms per one-batch step instead of ~4). Its copies are small and
        // follow the step they were issued behind.
        if (!c->partitioned) HZ_HIP(hipStreamCreateWithFlags(&c->s_copy, hipStreamNonBlocking));
        HZ_HIP(hipEventCreateWithFlags(&c->ev_staged, hipEventDisableTiming));
        HZ_HIP(hipEventCreateWithFlags(&c->ev_unpacked, hipEventDisableTiming));
        HZ_HIP(hipEventRecord(c->ev_unpacked, c->partitioned ? c->s_main : c->s_copy));
        c->staged.assign(c->lo.n_inst, 0);
    }
    hipStream_t s = stream ? (hipStream_t)stream : (c->partitioned ? c->s_main : c->s_copy);
    if (c->any_staged && c->stage_stream != s)   // ev_staged is one event: a second stream's copies would not be covered by it
        return set_err(HZ_ERR_ARG, "hz_inputs_stage: all stage calls between two enqueues of a context must use the same stream");
    c->stage_stream = s;
    HZ_HIP(hipStreamWaitEvent(s, c->ev_unpacked, 0));   // the slot's previous content has been scattered
    HZ_HIP(hipMemcpyAsync(slot, packed, bytes, hipMemcpyDefault, s));   // (`packed` may be device memory: inputs kept resident in packed form)
    HZ_HIP(hipEventRecord(c->ev_staged, s));   // all stage calls between two enqueues use one stream: the last record covers them
    c->staged[instance] = 1;
    c->any_staged = true;
    std::fill(c->input_set.begin(), c->input_set.end(), 1);
    return HZ_OK;
}

// `count` consecutive instances starting at `first`, instance first + j from packed + j * stride: what a serving loop stages per
// step. Contiguous host buffers (stride == bytes_each == hz_inputs_packed_bytes) cross PCIe as ONE copy -- the staging slots of
// consecutive instances are contiguous too -- instead of `count` calls and copies (32 per step and context in bench.py: 2 ms of
// submission time on the host thread that also has to enqueue the other context's step).
extern "C" hz_status hz_inputs_stage_range(hz_ctx* c, int32_t first, int32_t count, const void* packed, size_t bytes_each, size_t stride, void* stream) {
    if (count <= 0) return set_err(HZ_ERR_ARG, "hz_inputs_stage_range: count %d", count);
    if (stride < bytes_each) return set_err(HZ_ERR_ARG, "hz_inputs_stage_range: stride %zu < %zu bytes per instance", stride, bytes_each);
    uint8_t* slot = nullptr;
    hz_status st = upload_common(c, "hz_inputs_stage_range", first, packed, bytes_each, &slot);
    if (st != HZ_OK) return st;
    if ((uint32_t)first + (uint32_t)count > c->lo.n_inst)
        return set_err(HZ_ERR_ARG, "hz_inputs_stage_range: instances %d..%d out of range (n_instances = %u)", first, first + count - 1, c->lo.n_inst);
    if (stride != bytes_each) {
        for (int32_t j = 0; j < count; j++) {
            st = hz_inputs_stage(c, first + j, (const uint8_t*)packed + (size_t)j * stride, bytes_each, stream);
            if (st != HZ_OK) return st;
        }
        return HZ_OK;
    }
    st = hz_inputs_stage(c, first, packed, bytes_each, stream);   // creates the copy stream and events on first use; instance `first`
    if (st != HZ_OK || count == 1) return st;
    hipStream_t s = stream ? (hipStream_t)stream : (c->partitioned ? c->s_main : c->s_copy);
    HZ_HIP(hipMemcpyAsync(slot + bytes_each, (const uint8_t*)packed + bytes_each, (size_t)(count - 1) * bytes_each, hipMemcpyDefault, s));
    HZ_HIP(hipEventRecord(c->ev_staged, s));
    for (int32_t j = 1; j < count; j++) c->staged[first + j] = 1;
    return HZ_OK;
}

// ---- kernel schedule ---------------------------------------------------------------------------------
static SmtProcDesc make_proc(const SmtProcOff& o, uint32_t siblings, int which /*0: p1, 1: p2, 2: fee, 3: SMTProcessor main*/) {
    SmtProcDesc d;
    d.o = o;
    d.siblings = siblings;
    if (which == 1) {
        d.sc_oldkey = SC_KEY_S2OLD; d.sc_newkey = SC_KEY_2; d.sc_fnc0 = SC_P2_FNC0; d.sc_fnc1 = SC_P2_FNC1; d.sc_isold0 = SC_ISOLD0_2;
        d.sc_leaf_old = SC_LEAF_P2OLD; d.sc_leaf_new = SC_LEAF_P2NEW; d.sc_root_old = SC_ROOT_P2OLD; d.sc_root_new = SC_ROOT_P2NEW;
        d.cid_alias_old = C_RTX_P2_ALIAS_OLD; d.cid_alias_new = C_RTX_P2_ALIAS_NEW; d.cid_levins = C_RTX_P2_LEVINS; d.cid_sm_final = C_RTX_P2_SM_FINAL;
    } else {
        d.sc_oldkey = SC_KEY_S1OLD; d.sc_newkey = SC_KEY_1; d.sc_fnc0 = SC_P1_FNC0; d.sc_fnc1 = SC_P1_FNC1; d.sc_isold0 = SC_ISOLD0_1;
        d.sc_leaf_old = SC_LEAF_P1OLD; d.sc_leaf_new = SC_LEAF_P1NEW; d.sc_root_old = SC_ROOT_P1OLD; d.sc_root_new = SC_ROOT_P1NEW;
        if (which == 0) {
            d.cid_alias_old = C_RTX_P1_ALIAS_OLD; d.cid_alias_new = C_RTX_P1_ALIAS_NEW; d.cid_levins = C_RTX_P1_LEVINS; d.cid_sm_final = C_RTX_P1_SM_FINAL;
        } else if (which == 3) {
            d.cid_alias_old = C_SMTP_ALIAS_OLD; d.cid_alias_new = C_SMTP_ALIAS_NEW; d.cid_levins = C_SMTP_LEVINS; d.cid_sm_final = C_SMTP_SM_FINAL;
        } else {
            d.cid_alias_old = C_FEE_P_ALIAS_OLD; d.cid_alias_new = C_FEE_P_ALIAS_NEW; d.cid_levins = C_FEE_P_LEVINS; d.cid_sm_final = C_FEE_P_SM_FINAL;
        }
    }
    return d;
}

static Hash4Args make_hash4_rtx(uint8_t* base, Fr* sc, uint32_t n_units, const RtxOff& r) {
    Hash4Args h;
    memset(&h, 0, sizeof h);
    h.base = base; h.scratch = sc; h.n_units = n_units; h.n_jobs = 4;
    const PoseidonOff hs[4] = {r.oldSt1Hash, r.oldSt2Hash, r.newSt1Hash, r.newSt2Hash};
    const PoseidonOff h1[4] = {r.p1.hash1Old, r.p2.hash1Old, r.p1.hash1New, r.p2.hash1New};
    const uint32_t key[4] = {SC_KEY_S1OLD, SC_KEY_S2OLD, SC_KEY_1, SC_KEY_2};
    const uint32_t leaf[4] = {SC_LEAF_P1OLD, SC_LEAF_P2OLD, SC_LEAF_P1NEW, SC_LEAF_P2NEW};
    for (int j = 0; j < 4; j++) {
        HashJob& J = h.job[j];
        J.hs = hs[j]; J.h1 = h1[j]; J.sc_in = SC_HS_IN + 4 * j; J.sc_key = key[j]; J.sc_leaf = leaf[j];
        J.mux_off = ~0u; J.sc_ins = 0; J.sc_oldvalue = 0; J.out_sig = ~0u;
    }
    h.job[0].mux_off = r.mux16 + MX_S1OLDVALUE; h.job[0].sc_ins = SC_ISP1INSERT; h.job[0].sc_oldvalue = SC_OLDVALUE1;
    h.job[1].mux_off = r.mux16 + MX_S2OLDVALUE; h.job[1].sc_ins = SC_ISP2INSERT; h.job[1].sc_oldvalue = SC_OLDVALUE2;
    return h;
}

// the SMT chain kernel: one launch, one profile entry
static hipError_t enqueue_smt_chain(hz_ctx* c, const SmtArgs& sa0, const char* name, hipStream_t s) {
    SmtArgs sa = sa0;
    bool lat = c->smt_lat;
    if (!lat && c->smt_lat_few && c->partitioned && c->device >= 0 && c->device < 16) {
        MaskedPool& P = masked_pool();
        std::lock_guard<std::mutex> g(P.mu);
        lat = P.alive[c->device] <= 2;
    }
    sa.pos3_dense = lat ? (const Fr*)c->pos3.p : nullptr;
    const bool fee = sa.scratch == (Fr*)c->sc_fee.p;
    sa.zmark = (uint16_t*)(fee ? c->zm_fee.p : c->zm_tx.p);
    sa.skipped = (c->profiling && c->zm_skip.p) ? (unsigned long long*)c->zm_skip.p + (fee ? 1 : 0) : nullptr;
    if (!fee && getenv("HZ_SMT_TRACE")) {   // tools/experiments/smt_trace.py
        const uint32_t nl = sa.ucnt ? sa.ucnt : sa.n_units;
        const size_t waves = (size_t)((nl + 127) / 128) * 2 * 2 * sa.n_proc;
        if (c->smt_trace.bytes < waves * 32) { if (c->smt_trace.alloc(waves * 32) != hipSuccess) return hipErrorOutOfMemory; }
        (void)hipMemsetAsync(c->smt_trace.p, 0, waves * 32, s);
        sa.trace = (unsigned long long*)c->smt_trace.p;
    }
    ProfScope ps(c, s, name, sa.n_units);
    return launch_smt(sa, s);
}

static HashInputsArgs make_hi(hz_ctx* c, bool is_main);

// early_tail (RollupMain, whole batch): HashInputs does not wait for the SMT chains of every transaction. Its SHA-256 message needs
// the data-availability bits (front kernel), the last fee transaction's root (fee chain, own stream from the start) and ONE value
// of the chains: the LAST transaction's exit root. That transaction's four chains are evaluated first, as a launch of their own
// (4 lanes per batch) on the fee stream right after the hash-state kernel, then the message, the sequential chain and the
// expansion follow there while the main stream hashes the other 2047 transactions of every batch (its chain and back kernels leave
// the last one out: `skip_mod`, so no signal is written twice and a failing constraint of that transaction is recorded once).
// The 3.6 ms of one-wavefront chain latency and the expansion leave the end of the step.
static hz_status enqueue_rtx_tail(hz_ctx* c, uint8_t* base, uint32_t n_units, bool is_main, uint32_t sib1, uint32_t sib2, hipStream_t s, bool early_tail = false,
                                  bool early_prep = false, const MainFrontArgs* feeacc = nullptr) {
    const Layout& lo = c->lo;
    Fr* sc = (Fr*)c->sc_tx.p;
    ErrBuf* err = (ErrBuf*)c->err.p;
    EddsaArgs ea;
    memset(&ea, 0, sizeof ea);
    ea.base = base; ea.scratch = sc; ea.err = err; ea.n_units = n_units; ea.upi = lo.sections[lo.sec_tx].upi; ea.ed = lo.rtx.ed;
    const uint32_t u0 = is_main ? c->sh_first : 0, ucnt = is_main ? c->sh_count : 0;
    ea.u0 = u0; ea.ucnt = ucnt;
    ea.in_onChain = ~0u;
    if (is_main && !getenv("HZ_ED_NO_PRE_SPLIT")) {
        const auto& mi = lo.mi;
        ea.in_onChain = mi.onChain; ea.in_newAccount = mi.newAccount; ea.in_auxFromIdx = mi.auxFromIdx; ea.in_sign1 = mi.sign1; ea.in_ay1 = mi.ay1;
        ea.in_txCompressedData = mi.txCompressedData; ea.in_fromBjjCompressed = mi.fromBjjCompressed;
    }
    {
        // the split form (launches of <= HZ_ED_SPLIT_MAX signatures) parks its numerators in a side buffer: 84.7 KB per signature,
        // allocated by the first launch that takes that form and sized for it (a shard: its own range, not the section); the throughput
        // form keeps the lane state of the fixed-base kernel there (1.7 KB per eight signatures)
        const size_t need = eddsa_side_bytes(ucnt ? ucnt : n_units);
        if (need > c->ed_side.bytes || (need && !c->ed_side.p)) {
            HZ_HIP(hipStreamSynchronize(c->s_ed));
            HZ_HIP(hipStreamSynchronize(c->exclusive ? c->s_ed : c->s_fix));
            HZ_HIP(c->ed_side.alloc(need));
        }
    }
    ea.side = (uint32_t*)c->ed_side.p;
    {
        // the signature ladders only need the front step: run them beside the hash/SMT chain
        // the two halves of the check (S*B8 and R8 + h*8A) are independent: two kernels on two streams, then the equality
        hipStream_t sfix = c->exclusive ? c->s_ed : c->s_fix;
        HZ_HIP(hipStreamWaitEvent(sfix, c->ev_front, 0));
        { ProfScope ps(c, sfix, "eddsa_fix", n_units); HZ_HIP(launch_eddsa_fix(ea, sfix)); }
        // RollupMain's fee accumulators (k_main_feeacc) need the front step only and feed nothing downstream: they follow the fixed-base
        // half on its stream -- 10 + 1.5 ms against the 16.6 ms of the ladder beside it (one batch: 4.5 + 0.6 against 6.8) -- and are
        // joined with it (ev_fix -> the signature stream -> the launch stream)
        if (feeacc) { ProfScope ps(c, sfix, "fee_acc", n_units); HZ_HIP(launch_main_feeacc(*feeacc, sfix)); }
        HZ_HIP(hipEventRecord(c->ev_fix, sfix));
        // the signature stream waits for the front step INSIDE launch_eddsa: a RollupMain launch small enough for the split form starts
        // the point half of its prologue before that (k_eddsa_pre_a reads inputs only; ev_reset: the error buffer, the inputs' scatter)
        HZ_HIP(hipStreamWaitEvent(c->s_ed, c->ev_reset, 0));
        // (RollupMain: DecodeTx's sigL2Hash comes from k_main_sighash at the head of the fee stream -- ev_sighash; the hash half of the
        // prologue waits for it, the point half and the ladder's head do not)
        { ProfScope ps(c, c->s_ed, "eddsa", n_units); HZ_HIP(launch_eddsa(ea, c->s_ed, c->ev_front, feeacc ? c->ev_sighash : nullptr)); }
        HZ_HIP(hipStreamWaitEvent(c->s_ed, c->ev_fix, 0));
        { ProfScope ps(c, c->s_ed, "eddsa_final", n_units); HZ_HIP(launch_eddsa_final(ea, c->s_ed)); }
        HZ_HIP(hipEventRecord(c->ev_ed, c->s_ed));
    }
    {
        Hash4Args h4 = make_hash4_rtx(base, sc, n_units, lo.rtx);
        h4.u0 = u0; h4.ucnt = ucnt;
        ProfScope ps(c, s, "hash4", n_units);
        HZ_HIP(launch_hash4(h4, s));
    }
    SmtArgs sa;
    memset(&sa, 0, sizeof sa);
    sa.base = base; sa.scratch = sc; sa.err = err; sa.n_units = n_units; sa.n_levels = (uint32_t)lo.p.L + 1; sa.n_proc = 2; sa.upi = lo.sections[lo.sec_tx].upi;
    sa.p[0] = make_proc(lo.rtx.p1, sib1, 0);
    sa.p[1] = make_proc(lo.rtx.p2, sib2, 1);
    sa.u0 = u0; sa.ucnt = ucnt;
    RtxBackArgs ba;
    memset(&ba, 0, sizeof ba);
    ba.base = base; ba.glob_base = is_main ? sec_ptr(c, lo.sec_glob) : nullptr; ba.scratch = sc; ba.err = err; ba.n_units = n_units; ba.L = (uint32_t)lo.p.L;
    ba.is_main = is_main ? 1 : 0;
    ba.upi = lo.sections[lo.sec_tx].upi; ba.B = lo.n_inst;
    ba.u0 = u0; ba.ucnt = ucnt;
    ba.p[0] = sa.p[0]; ba.p[1] = sa.p[1];
    ba.s3 = lo.rtx.s3; ba.s4 = lo.rtx.s4; ba.s5 = lo.rtx.s5;
    if (is_main) {
        ba.im_stateroot = lo.mi.imStateRoot; ba.im_exitroot = lo.mi.imExitRoot; ba.g_initfeeroot = lo.g.imInitStateRootFee;
        ba.main_l1l2amt = lo.rtx.main_l1l2amt; ba.n2bAmount = lo.dec.n2bAmount;
    } else {
        ba.o_newStateRoot = lo.rtxi.o_newStateRoot; ba.o_newExitRoot = lo.rtxi.o_newExitRoot;
    }
    if (early_tail) {
        hipStream_t st = c->s_fee;   // the fee chain was enqueued on it before: newStateRoot is ready when these run
        HZ_HIP(hipEventRecord(c->ev_hash4, s));
        HZ_HIP(hipStreamWaitEvent(st, c->ev_hash4, 0));
        SmtArgs sl = sa;
        sl.u0 = (uint32_t)lo.p.nTx - 1; sl.ucnt = lo.n_inst; sl.ustride = (uint32_t)lo.p.nTx;
        sl.p[0].sc_root_old = SC_EROOT_P1OLD; sl.p[0].sc_root_new = SC_EROOT_P1NEW; sl.p[1].sc_root_old = SC_EROOT_P2OLD; sl.p[1].sc_root_new = SC_EROOT_P2NEW;
        sl.zmark = (uint16_t*)c->zm_tx.p;   // (marks are per unit: these units are nobody else's this step)
        sl.skipped = (c->profiling && c->zm_skip.p) ? (unsigned long long*)c->zm_skip.p + 2 : nullptr;
        HZ_HIP(launch_smt(sl, st));
        RtxBackArgs bl = ba;
        bl.u0 = sl.u0; bl.ucnt = sl.ucnt; bl.ustride = sl.ustride;
        bl.p[0] = sl.p[0]; bl.p[1] = sl.p[1];
        bl.skip_h = 1;
        HZ_HIP(launch_rtx_back(bl, st));
        HZ_HIP(launch_da_mask(ba, st));
        // the main stream leaves these units (and phase H of every unit: k_da_mask has it) to the launches above
        sa.skip_mod = ba.skip_mod = (uint32_t)lo.p.nTx;
        ba.skip_h = 1;
        {
            ProfScope ps(c, st, "hash_inputs", (uint64_t)lo.hi.sha.nblocks);
            HZ_HIP(launch_hash_inputs(make_hi(c, true), st, c->exclusive ? nullptr : c->s_sha, c->ev_sha, 9));
        }
        HZ_HIP(hipEventRecord(c->ev_tail, st));
    }
    if (early_prep) {
        // CU-partitioned contexts (a few batches, latency): the chain cannot start before the roots are known (they sit in the first
        // block), but the rest of the message can be laid out while the SMT chains run
        hipStream_t st = c->s_fee;
        HZ_HIP(hipStreamWaitEvent(st, c->ev_front, 0));
        HZ_HIP(launch_da_mask(ba, st));
        ba.skip_h = 1;
        HZ_HIP(launch_hi_prep_body(make_hi(c, true), st));
        HZ_HIP(hipEventRecord(c->ev_tail, st));
    }
    HZ_HIP(enqueue_smt_chain(c, sa, "smt", s));
    { ProfScope ps(c, s, "rtx_back", n_units); HZ_HIP(launch_rtx_back(ba, s)); }
    return HZ_OK;   // the caller joins the signature stream (ev_ed) after whatever else it launches on `s`
}

static hz_status enqueue_fee(hz_ctx* c, uint8_t* base, uint32_t n_units, bool is_main, hipStream_t s) {
    const Layout& lo = c->lo;
    Fr* sc = (Fr*)c->sc_fee.p;
    ErrBuf* err = (ErrBuf*)c->err.p;
    FeeFrontArgs fa;
    memset(&fa, 0, sizeof fa);
    fa.base = base; fa.glob_base = is_main ? sec_ptr(c, lo.sec_glob) : nullptr; fa.scratch = sc; fa.err = err; fa.n_units = n_units; fa.is_main = is_main;
    fa.upi = lo.sections[lo.sec_fee].upi; fa.B = lo.n_inst;
    fa.fee = lo.fee;
    uint32_t sib;
    if (is_main) {
        const MainFeeInOff& f = lo.fi;
        fa.in_feePlanToken = f.feePlanTokens; fa.in_feeIdx = f.feeIdxs; fa.in_accFee = f.imFinalAccFee; fa.in_tokenID = f.tokenID3; fa.in_nonce = f.nonce3;
        fa.in_sign = f.sign3; fa.in_balance = f.balance3; fa.in_ay = f.ay3; fa.in_ethAddr = f.ethAddr3; fa.im_stateRootFee = f.imStateRootFee;
        fa.g_initfeeroot = lo.g.imInitStateRootFee;
        sib = f.siblings3;
    } else {
        const FeeTxInOff& f = lo.feei;
        fa.in_feePlanToken = f.feePlanToken; fa.in_feeIdx = f.feeIdx; fa.in_accFee = f.accFee; fa.in_tokenID = f.tokenID; fa.in_nonce = f.nonce;
        fa.in_sign = f.sign; fa.in_balance = f.balance; fa.in_ay = f.ay; fa.in_ethAddr = f.ethAddr; fa.in_oldStateRoot = f.oldStateRoot;
        sib = f.siblings;
    }
    { ProfScope ps(c, s, "fee_front", n_units); HZ_HIP(launch_fee_front(fa, s)); }
    Hash4Args h;
    memset(&h, 0, sizeof h);
    h.base = base; h.scratch = sc; h.n_units = n_units; h.n_jobs = 2;
    h.job[0] = HashJob{lo.fee.oldHash, lo.fee.p.hash1Old, SC_HS_IN + 0, SC_KEY_S1OLD, SC_LEAF_P1OLD, ~0u, 0, 0, ~0u};
    h.job[1] = HashJob{lo.fee.newHash, lo.fee.p.hash1New, SC_HS_IN + 8, SC_KEY_1, SC_LEAF_P1NEW, ~0u, 0, 0, ~0u};
    { ProfScope ps(c, s, "fee_hash", n_units); HZ_HIP(launch_hash4(h, s)); }
    SmtArgs sa;
    memset(&sa, 0, sizeof sa);
    sa.base = base; sa.scratch = sc; sa.err = err; sa.n_units = n_units; sa.n_levels = (uint32_t)lo.p.L + 1; sa.n_proc = 1; sa.upi = lo.sections[lo.sec_fee].upi;
    sa.p[0] = make_proc(lo.fee.p, sib, 2);
    HZ_HIP(enqueue_smt_chain(c, sa, "fee_smt", s));
    FeeBackArgs fb;
    memset(&fb, 0, sizeof fb);
    fb.base = base; fb.scratch = sc; fb.err = err; fb.n_units = n_units; fb.is_main = is_main; fb.upi = lo.sections[lo.sec_fee].upi; fb.p = sa.p[0];
    fb.im_stateRootFee = is_main ? lo.fi.imStateRootFee : 0; fb.o_newStateRoot = lo.fee.o_newStateRoot;
    { ProfScope ps(c, s, "fee_back", n_units); HZ_HIP(launch_fee_back(fb, s)); }
    return HZ_OK;
}

static HashInputsArgs make_hi(hz_ctx* c, bool is_main) {
    const Layout& lo = c->lo;
    HashInputsArgs a;
    memset(&a, 0, sizeof a);
    a.hi_base = sec_ptr(c, lo.sec_hi);
    a.err = (ErrBuf*)c->err.p;
    a.nTx = (uint32_t)lo.p.nTx; a.L = (uint32_t)lo.p.L; a.maxL1 = (uint32_t)lo.p.maxL1; a.F = (uint32_t)lo.p.F; a.is_main = is_main; a.B = lo.n_inst;
    a.hi = lo.hi;
    a.msg = (uint8_t*)c->msg.p; a.chain = (uint32_t*)c->chain.p;
    if (is_main) {
        a.glob_base = sec_ptr(c, lo.sec_glob); a.tx_base = sec_ptr(c, lo.sec_tx); a.fee_base = sec_ptr(c, lo.sec_fee);
        a.tx_scratch = (Fr*)c->sc_tx.p; a.fee_scratch = (Fr*)c->sc_fee.p;
        a.g = lo.g; a.mi_onChain = lo.mi.onChain; a.fi_feeIdxs = lo.fi.feeIdxs; a.dec = lo.dec; a.rtx_s5 = lo.rtx.s5; a.rtx_main_l1l2amt = lo.rtx.main_l1l2amt;
    }
    return a;
}

// error buffer header: minkey = ~0, filter (= ~0 unless this is a second pass), count = 0
static hz_status reset_err(hz_ctx* c, hipStream_t s, unsigned long long filter) {
    HZ_HIP(hipMemsetAsync(c->err.p, 0xFF, 16, s));
    HZ_HIP(hipMemsetAsync((uint8_t*)c->err.p + 16, 0, 8, s));
    if (filter != ~0ull) {
        c->filter_stage = filter;
        HZ_HIP(hipMemcpyAsync((uint8_t*)c->err.p + 8, &c->filter_stage, 8, hipMemcpyHostToDevice, s));
    } else {
        // the per-instance minima are reset only when the last pass lowered one (or nobody checked): a clean serving loop pays nothing
        if (c->inst_min_dirty || !c->checked) HZ_HIP(hipMemsetAsync(c->inst_min.p, 0xFF, c->inst_min.bytes, s));
        c->inst_min_dirty = false;
        c->checked = false;
    }
    return HZ_OK;
}

// filter != ~0: a second pass over the SAME inputs for the operands of failures the first pass found (hz_witness_check after an
// overflow of the record list; hz_witness_failures): inputs staged for the next step stay staged, the per-instance minima stay.
static hz_status enqueue_impl(hz_ctx* c, void* stream, unsigned long long filter) {
    const bool rerun = filter != ~0ull;
    if (!c) return set_err(HZ_ERR_ARG, "hz_witness_enqueue: null context");
    const Layout& lo = c->lo;
    for (size_t i = 0; i < c->input_set.size(); i++)
        if (!c->input_set[i]) return set_err(HZ_ERR_INPUT, "Not all inputs have been set: %s", lo.inputs[i].name.c_str());
    HZ_HIP(hipSetDevice(c->device));
    // the legacy default stream has implicit-synchronisation semantics that do not mix with the
    // context's non-blocking side streams: a NULL stream means "the context's own stream"
    hipStream_t s = stream ? (hipStream_t)stream : c->s_main;
    if (c->any_staged && !rerun) {   // scatter the staged inputs (hz_inputs_stage) into the witness layout
        HZ_HIP(hipStreamWaitEvent(s, c->ev_staged, 0));
        for (uint32_t b = 0; b < lo.n_inst;) {   // one launch per run of consecutive staged instances (usually: all of them)
            if (!c->staged[b]) { b++; continue; }
            uint32_t e = b;
            while (e < lo.n_inst && c->staged[e] && e - b < 65535u) c->staged[e++] = 0;
            const hz_status st = launch_unpack(c, b, s, e - b);
            if (st != HZ_OK) return st;
            b = e;
        }
        HZ_HIP(hipEventRecord(c->ev_unpacked, s));
        c->any_staged = false;
    }
    if (c->inputs_pending) {
        HZ_HIP(hipStreamWaitEvent(s, c->ev_inputs, 0));
        c->inputs_pending = false;
    }
    // partitioned contexts run their main sequence on the CU-masked stream, ordered after / before the caller's stream by two events
    hipStream_t s_user = s;
    if (c->partitioned && s != c->s_main) {
        HZ_HIP(hipEventRecord(c->ev_user_in, s_user));
        s = c->s_main;
        HZ_HIP(hipStreamWaitEvent(s, c->ev_user_in, 0));
    }
    // profiling mode 2: every kernel alone on the device (the side streams alias the launch stream)
    struct StreamAlias {
        hz_ctx* c; hipStream_t ed, fee;
        StreamAlias(hz_ctx* c_, hipStream_t s_) : c(c_), ed(c_->s_ed), fee(c_->s_fee) { if (c->exclusive) c->s_ed = c->s_fee = s_; }
        ~StreamAlias() { c->s_ed = ed; c->s_fee = fee; }
    } alias(c, s);
    { const hz_status st = reset_err(c, s, filter); if (st != HZ_OK) return st; }
    if (c->zm_reset) {   // (before ev_reset: every side stream waits for it before its first kernel)
        if (c->zm_tx.p) HZ_HIP(hipMemsetAsync(c->zm_tx.p, 0xFF, c->zm_tx.bytes, s));
        if (c->zm_fee.p) HZ_HIP(hipMemsetAsync(c->zm_fee.p, 0xFF, c->zm_fee.bytes, s));
        c->zm_reset = false;
    }
    if (c->profiling && c->zm_skip.p) HZ_HIP(hipMemsetAsync(c->zm_skip.p, 0, c->zm_skip.bytes, s));
    HZ_HIP(hipEventRecord(c->ev_reset, s));
    ErrBuf* err = (ErrBuf*)c->err.p;
    c->prof_used = 0;
    switch (lo.p.tmpl) {
        case T_ROLLUP_MAIN: {
            MainFrontArgs fa;
            memset(&fa, 0, sizeof fa);
            fa.tx_base = sec_ptr(c, lo.sec_tx); fa.fee_base = sec_ptr(c, lo.sec_fee); fa.glob_base = sec_ptr(c, lo.sec_glob);
            fa.scratch = (Fr*)c->sc_tx.p; fa.err = err; fa.nTx = (uint32_t)lo.p.nTx; fa.L = (uint32_t)lo.p.L; fa.F = (uint32_t)lo.p.F; fa.B = lo.n_inst;
            fa.g = lo.g; fa.mi = lo.mi; fa.fi = lo.fi; fa.dec = lo.dec; fa.rtx = lo.rtx;
            hz_status st = HZ_OK;
            const bool tail_now = !c->sharded;   // sharded contexts run the tail separately (hz_witness_enqueue_tail)
            if (c->sharded && c->sh_count == 0) break;   // an empty shard (more ranks than transactions): nothing to evaluate
            fa.u0 = c->sh_first; fa.ucnt = c->sh_count;
            // DecodeTx's sigL2Hash (k_main_sighash: one Poseidon of width 7 over INPUTS; read by the signature prologue's hash half only)
            // heads the fee stream -- beside the front kernel and the prologue's point half, in front of a chain that has time to spare
            HZ_HIP(hipStreamWaitEvent(c->s_fee, c->ev_reset, 0));
            { ProfScope ps(c, c->s_fee, "sig_hash", (uint64_t)fa.nTx * fa.B); HZ_HIP(launch_main_sighash(fa, c->s_fee)); }
            HZ_HIP(hipEventRecord(c->ev_sighash, c->s_fee));
            if (tail_now) {
                st = enqueue_fee(c, fa.fee_base, lo.sections[lo.sec_fee].n_units, true, c->s_fee);   // independent of the transactions
                if (st != HZ_OK) return st;
                HZ_HIP(hipEventRecord(c->ev_fee, c->s_fee));
            }
            fa.u0 = c->sh_first; fa.ucnt = c->sh_count;
            { ProfScope ps(c, s, "front", (uint64_t)fa.nTx * fa.B); HZ_HIP(launch_main_front(fa, s)); }
            HZ_HIP(hipEventRecord(c->ev_front, s));
            const bool early = tail_now && !c->partitioned;   // CU-partitioned contexts keep the tail on the main stream
            const bool early_prep = tail_now && c->partitioned;
            st = enqueue_rtx_tail(c, fa.tx_base, lo.sections[lo.sec_tx].n_units, true, lo.mi.siblings1, lo.mi.siblings2, s, early, early_prep, &fa);
            if (st != HZ_OK) return st;
            if (early) {
                HZ_HIP(hipStreamWaitEvent(s, c->ev_tail, 0));
            } else if (tail_now) {
                // HashInputs needs the roots and the data-availability bits, not the signatures: it runs beside the ladders
                // (partitioned contexts: chain and expansion on this stream, one launch each -- the expansion piped over the fee stream
                // in eight groups, as the throughput schedule does, costs a single batch 1.25 ms: 9.1 against 7.86 ms)
                HZ_HIP(hipStreamWaitEvent(s, c->ev_fee, 0));
                if (early_prep) HZ_HIP(hipStreamWaitEvent(s, c->ev_tail, 0));
                { ProfScope ps(c, s, "hash_inputs", (uint64_t)lo.hi.sha.nblocks); HZ_HIP(launch_hash_inputs(make_hi(c, true), s, c->partitioned ? nullptr : c->s_fee, c->ev_sha, 9, early_prep)); }
            }
            HZ_HIP(hipStreamWaitEvent(s, c->ev_ed, 0));   // join the signature stream
            break;
        }
        case T_ROLLUP_TX: {
            RtxFrontArgs fa;
            memset(&fa, 0, sizeof fa);
            fa.base = sec_ptr(c, 0); fa.scratch = (Fr*)c->sc_tx.p; fa.err = err; fa.N = lo.sections[0].n_units; fa.L = (uint32_t)lo.p.L; fa.F = (uint32_t)lo.p.F;
            fa.in = lo.rtxi; fa.rtx = lo.rtx;
            { ProfScope ps(c, s, "front", fa.N); HZ_HIP(launch_rtx_front(fa, s)); }
            HZ_HIP(hipEventRecord(c->ev_front, s));
            hz_status st = enqueue_rtx_tail(c, fa.base, fa.N, false, lo.rtxi.siblings1, lo.rtxi.siblings2, s);
            if (st != HZ_OK) return st;
            HZ_HIP(hipStreamWaitEvent(s, c->ev_ed, 0));   // join the signature stream
            break;
        }
        case T_DECODE_TX: {
            DecMainArgs da;
            memset(&da, 0, sizeof da);
            da.base = sec_ptr(c, 0); da.err = err; da.N = lo.sections[0].n_units; da.L = (uint32_t)lo.p.L; da.in = lo.deci; da.dec = lo.dec;
            HZ_HIP(launch_dec_main(da, s));
            break;
        }
        case T_FEE_TX: {
            hz_status st = enqueue_fee(c, sec_ptr(c, 0), lo.sections[0].n_units, false, s);
            if (st != HZ_OK) return st;
            break;
        }
        case T_HASH_STATE:
            HZ_HIP(launch_hash_state_main(sec_ptr(c, 0), lo.sections[0].n_units, lo.hs, s));
            break;
        case T_WITHDRAW: {
            WithdrawArgs wa;
            memset(&wa, 0, sizeof wa);
            wa.base = sec_ptr(c, 0); wa.err = err; wa.N = lo.sections[0].n_units; wa.L = (uint32_t)lo.p.L; wa.wd = lo.wd;
            // the SHA-256 bit witness (store bound) runs beside the Poseidon / SMT part (integer bound) on a side stream
            HZ_HIP(hipStreamWaitEvent(c->s_ed, c->ev_reset, 0));
            { ProfScope ps(c, c->s_ed, "withdraw_sha", wa.N); HZ_HIP(launch_withdraw_sha(wa, c->s_ed)); }
            HZ_HIP(hipEventRecord(c->ev_ed, c->s_ed));
            { ProfScope ps(c, s, "withdraw", wa.N); HZ_HIP(launch_withdraw(wa, s)); }
            HZ_HIP(hipStreamWaitEvent(s, c->ev_ed, 0));
            break;
        }
        case T_HASH_INPUTS:
            HZ_HIP(launch_hash_inputs(make_hi(c, false), s, c->s_fee, c->ev_sha, 9));
            break;
        case T_SMT_PROCESSOR: case T_SMT_VERIFIER: {
            SmtMainArgs ma;
            memset(&ma, 0, sizeof ma);
            ma.base = sec_ptr(c, 0); ma.scratch = (Fr*)c->sc_fee.p; ma.err = err; ma.N = lo.sections[0].n_units; ma.n_levels = (uint32_t)lo.p.L;
            if (lo.p.tmpl == T_SMT_VERIFIER) {
                ma.vin = lo.smtvi; ma.ver = lo.smtv;
                HZ_HIP(launch_smtver_main(ma, s));
                break;
            }
            ma.pin = lo.smtpi; ma.proc = make_proc(lo.smtp, lo.smtpi.siblings, 3);
            HZ_HIP(launch_smtproc_front(ma, s));
            SmtArgs sa;
            memset(&sa, 0, sizeof sa);
            sa.base = ma.base; sa.scratch = ma.scratch; sa.err = err; sa.n_units = ma.N; sa.n_levels = ma.n_levels; sa.n_proc = 1; sa.upi = 1;
            sa.p[0] = ma.proc;
            HZ_HIP(enqueue_smt_chain(c, sa, "smt", s));
            HZ_HIP(launch_smtproc_back(ma, s));
            break;
        }
        case T_DECODE_FLOAT: case T_COMPUTE_FEE: case T_FEE_ACCUMULATOR: case T_BALANCE_UPDATER: case T_ROLLUP_TX_STATES: case T_RQ_TX_VERIFIER:
        case T_MUX256: case T_BITS2AYSIGN: case T_AYSIGN2AX: {
            GadgetArgs ga;
            memset(&ga, 0, sizeof ga);
            ga.base = sec_ptr(c, 0); ga.err = err; ga.N = lo.sections[0].n_units; ga.F = (uint32_t)lo.p.F; ga.io = lo.gad;
            ga.n2b40 = lo.rtx.n2bLoadAmountF; ga.df = lo.rtx.dfLoadAmount; ga.bu = lo.rtx.bu; ga.feeAcc = lo.rtx.feeAcc; ga.st = lo.rtx.st;
            ga.rq_n2b = lo.rtx.rq_n2b;
            for (int m = 0; m < 3; m++) ga.rq_mux[m] = lo.rtx.rq_mux[m];
            if (lo.p.tmpl == T_AYSIGN2AX) HZ_HIP(launch_ay_sign_2_ax_main(ga, lo.rtx.ed, s));
            else HZ_HIP(launch_gadget(lo.p.tmpl, ga, s));
            break;
        }
    }
    if (s != s_user) {
        HZ_HIP(hipEventRecord(c->ev_user_out, s));
        HZ_HIP(hipStreamWaitEvent(s_user, c->ev_user_out, 0));
    }
    c->last_stream = s_user;
    c->enqueued = true;
    return HZ_OK;
}

extern "C" hz_status hz_witness_enqueue(hz_ctx* c, void* stream) { return enqueue_impl(c, stream, ~0ull); }

static void fill_error(hz_error* out, unsigned long long key, const ErrRec* r) {
    if (!out) return;
    memset(out, 0, sizeof *out);
    out->instance = (int32_t)(key >> 40);
    out->unit = (int32_t)((key >> 16) & 0xFFFFFF);
    out->constraint_id = (int32_t)(key & 0xFFFF);
    if (r) {
        memcpy(out->lhs, r->lhs, 32);
        memcpy(out->rhs, r->rhs, 32);
    }
}

extern "C" hz_status hz_witness_check(hz_ctx* c, hz_error* out) {
    if (!c || !c->enqueued) return set_err(HZ_ERR_ARG, "hz_witness_check: nothing enqueued");
    HZ_HIP(hipSetDevice(c->device));
    struct { unsigned long long minkey, filter; unsigned int count, pad; } hd;
    for (int attempt = 0; attempt < 2; attempt++) {
        // copy on the launch stream: a default-stream hipMemcpy would wait for every other batch in flight
        unsigned long long bad = ~0ull;
        HZ_HIP(hipMemcpyAsync(&hd, c->err.p, sizeof hd, hipMemcpyDeviceToHost, c->last_stream));
        if (c->upl_used && attempt == 0) HZ_HIP(hipMemcpyAsync(&bad, c->upl_bad.p, sizeof bad, hipMemcpyDeviceToHost, c->last_stream));
        HZ_HIP(hipStreamSynchronize(c->last_stream));
        c->enqueued = false;
        if (bad != ~0ull) {   // an uploaded element was >= r: the witness computed from it means nothing
            HZ_HIP(hipMemsetAsync(c->upl_bad.p, 0xFF, sizeof bad, c->last_stream));
            const size_t idx = (size_t)(bad >> 40);
            return set_err(HZ_ERR_INPUT, "input %s[%llu] is not a canonical field element (>= r)", idx < c->lo.inputs.size() ? c->lo.inputs[idx].name.c_str() : "?",
                           (unsigned long long)(bad & ((1ull << 40) - 1)));
        }
        c->checked = true;
        c->last_minkey = hd.minkey;
        c->inst_min_dirty = hd.minkey != ~0ull;
        if (hd.minkey == ~0ull) return HZ_OK;
        const unsigned int n = std::min<unsigned int>(hd.count, HZ_ERR_CAP);
        std::vector<ErrRec> recs(n);
        if (n) {
            HZ_HIP(hipMemcpyAsync(recs.data(), (uint8_t*)c->err.p + offsetof(ErrBuf, rec), n * sizeof(ErrRec), hipMemcpyDeviceToHost, c->last_stream));
            HZ_HIP(hipStreamSynchronize(c->last_stream));
        }
        for (const ErrRec& r : recs)
            if (r.key == hd.minkey) {
                fill_error(out, hd.minkey, &r);
                return set_err(HZ_ERR_CONSTRAINT, "Constraint doesn't match (%s, instance %d unit %d)", constraint_name((int)(hd.minkey & 0xFFFF)),
                               (int)(hd.minkey >> 40), (int)((hd.minkey >> 16) & 0xFFFFFF));
            }
        if (attempt == 1) break;
        // the record list overflowed before the first failure was appended: run again, recording only it
        hz_status st = enqueue_impl(c, c->last_stream, hd.minkey);
        if (st != HZ_OK) return st;
    }
    fill_error(out, hd.minkey, nullptr);
    return set_err(HZ_ERR_CONSTRAINT, "Constraint doesn't match (%s, instance %d unit %d; operands not captured)", constraint_name((int)(hd.minkey & 0xFFFF)),
                   (int)(hd.minkey >> 40), (int)((hd.minkey >> 16) & 0xFFFFFF));
}

// The first violated constraint of EVERY instance of the last launch (a step evaluates 32 batches at once: the caller wants to know
// which of them to reject, not only the first). The first pass already left each instance's lowest key in inst_min; the operands
// come from one more pass over the same inputs in which the lane that owns an instance's lowest key writes them to the instance's slot.
extern "C" hz_status hz_witness_failures(hz_ctx* c, hz_error* out, size_t cap, size_t* n_failed) {
    if (!c || !n_failed || (cap && !out)) return set_err(HZ_ERR_ARG, "hz_witness_failures: null argument");
    if (!c->checked) return set_err(HZ_ERR_ARG, "hz_witness_failures: call hz_witness_check / hz_witness_run first");
    if (c->sharded) return set_err(HZ_ERR_ARG, "hz_witness_failures: not for tx-sharded contexts (one batch: hz_witness_check reports it)");
    *n_failed = 0;
    if (c->last_minkey == ~0ull) return HZ_OK;
    HZ_HIP(hipSetDevice(c->device));
    const uint32_t B = c->lo.n_inst;
    hipStream_t s = c->last_stream ? c->last_stream : c->s_main;
    std::vector<unsigned long long> mins(B);
    HZ_HIP(hipMemcpyAsync(mins.data(), c->inst_min.p, (size_t)B * 8, hipMemcpyDeviceToHost, s));
    HZ_HIP(hipStreamSynchronize(s));
    if (cap == 0) {   // count only: the first pass knows it
        for (uint32_t b = 0; b < B; b++) *n_failed += mins[b] != ~0ull;
        return HZ_OK;
    }
    if (!c->inst_rec.p) {
        HZ_HIP(c->inst_rec.alloc((size_t)B * sizeof(ErrRec)));
        HZ_HIP(hipMemcpyAsync((uint8_t*)c->err.p + offsetof(ErrBuf, inst_rec), &c->inst_rec.p, sizeof(void*), hipMemcpyHostToDevice, s));
    }
    HZ_HIP(hipMemsetAsync(c->inst_rec.p, 0xFF, c->inst_rec.bytes, s));
    hz_status st = enqueue_impl(c, s, HZ_FILTER_PER_INST);
    if (st != HZ_OK) return st;
    std::vector<ErrRec> recs(B);
    HZ_HIP(hipMemcpyAsync(recs.data(), c->inst_rec.p, (size_t)B * sizeof(ErrRec), hipMemcpyDeviceToHost, c->last_stream));
    HZ_HIP(hipStreamSynchronize(c->last_stream));
    c->enqueued = false;
    size_t k = 0;
    for (uint32_t b = 0; b < B; b++) {
        if (mins[b] == ~0ull) continue;
        if (k < cap) fill_error(&out[k], mins[b], recs[b].key == mins[b] ? &recs[b] : nullptr);
        k++;
    }
    *n_failed = k;
    return HZ_OK;
}

// ---- multi-GPU intra-batch sharding ------------------------------------------------------------------
extern "C" hz_status hz_ctx_set_shard(hz_ctx* c, int32_t first, int32_t count, int32_t tail) {
    if (!c || c->lo.p.tmpl != T_ROLLUP_MAIN) return set_err(HZ_ERR_ARG, "hz_ctx_set_shard: RollupMain contexts only");
    if (c->lo.n_inst != 1) return set_err(HZ_ERR_ARG, "hz_ctx_set_shard: one batch per context when sharding");
    if (count < 0) {   // back to the whole batch
        c->sharded = false; c->sh_first = c->sh_count = 0; c->sh_tail = true;
        return HZ_OK;
    }
    if (first < 0 || first > c->lo.p.nTx || count > c->lo.p.nTx - first) return set_err(HZ_ERR_ARG, "hz_ctx_set_shard: bad range");
    c->sharded = true;
    c->sh_first = (uint32_t)first;
    c->sh_count = (uint32_t)count;
    c->sh_tail = tail != 0;
    return HZ_OK;
}
extern "C" uint64_t hz_da_record_bytes(const hz_ctx*) { return HZ_DA_RECORD_BYTES; }
static DaArgs make_da(hz_ctx* c, uint32_t first, uint32_t count, void* buf) {
    const Layout& lo = c->lo;
    DaArgs a;
    memset(&a, 0, sizeof a);
    a.tx_base = sec_ptr(c, lo.sec_tx); a.tx_scratch = (Fr*)c->sc_tx.p; a.buf = (uint8_t*)buf;
    a.nTx = (uint32_t)lo.p.nTx; a.L = (uint32_t)lo.p.L; a.u0 = first; a.ucnt = count;
    a.l1full = lo.dec.l1full; a.n2bData = lo.dec.n2bData; a.n2bFinalToIdx = lo.dec.n2bFinalToIdx; a.l1l2amt = lo.rtx.main_l1l2amt;
    a.l1l2Fee = lo.dec.l1l2Fee; a.s5 = lo.rtx.s5;
    return a;
}
extern "C" hz_status hz_da_export(hz_ctx* c, void* d_buf, void* stream) {
    if (!c || c->lo.p.tmpl != T_ROLLUP_MAIN || !d_buf) return set_err(HZ_ERR_ARG, "hz_da_export: bad argument");
    HZ_HIP(hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->s_main;
    const uint32_t cnt = c->sharded ? c->sh_count : (uint32_t)c->lo.p.nTx;
    HZ_HIP(launch_da_export(make_da(c, c->sh_first, cnt, d_buf), s));
    return HZ_OK;
}
extern "C" hz_status hz_da_import(hz_ctx* c, int32_t first, int32_t count, const void* d_buf, void* stream) {
    if (!c || c->lo.p.tmpl != T_ROLLUP_MAIN || !d_buf || first < 0 || count < 0 || first > c->lo.p.nTx || count > c->lo.p.nTx - first)
        return set_err(HZ_ERR_ARG, "hz_da_import: bad argument");
    HZ_HIP(hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->s_main;
    HZ_HIP(launch_da_import(make_da(c, (uint32_t)first, (uint32_t)count, const_cast<void*>(d_buf)), s));
    return HZ_OK;
}
// fee transactions + HashInputs of a sharded context, after the other ranks' records were imported
extern "C" hz_status hz_witness_enqueue_tail(hz_ctx* c, void* stream) {
    if (!c || c->lo.p.tmpl != T_ROLLUP_MAIN) return set_err(HZ_ERR_ARG, "hz_witness_enqueue_tail: RollupMain contexts only");
    HZ_HIP(hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->s_main;
    const Layout& lo = c->lo;
    hz_status st = enqueue_fee(c, sec_ptr(c, lo.sec_fee), lo.sections[lo.sec_fee].n_units, true, s);
    if (st != HZ_OK) return st;
    { ProfScope ps(c, s, "hash_inputs", (uint64_t)lo.hi.sha.nblocks); HZ_HIP(launch_hash_inputs(make_hi(c, true), s, c->s_fee, c->ev_sha, 9)); }
    c->last_stream = s;
    c->enqueued = true;
    return HZ_OK;
}

extern "C" hz_status hz_witness_enqueue_tail_chain(hz_ctx* c, void* stream) {
    if (!c || c->lo.p.tmpl != T_ROLLUP_MAIN) return set_err(HZ_ERR_ARG, "hz_witness_enqueue_tail_chain: RollupMain contexts only");
    HZ_HIP(hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->s_main;
    const Layout& lo = c->lo;
    hz_status st = enqueue_fee(c, sec_ptr(c, lo.sec_fee), lo.sections[lo.sec_fee].n_units, true, s);
    if (st != HZ_OK) return st;
    { ProfScope ps(c, s, "hash_inputs", (uint64_t)lo.hi.sha.nblocks); HZ_HIP(launch_hi_chain_only(make_hi(c, true), s)); }
    c->last_stream = s;
    c->enqueued = true;
    return HZ_OK;
}
extern "C" uint64_t hz_sha_blocks(const hz_ctx* c) { return (c && c->lo.sec_hi >= 0) ? (uint64_t)c->lo.hi.sha.nblocks : 0; }
extern "C" uint64_t hz_sha_state_bytes(const hz_ctx* c) {
    return (c && c->lo.sec_hi >= 0) ? ((uint64_t)c->lo.hi.sha.nblocks * 64 + ((uint64_t)c->lo.hi.sha.nblocks + 1) * 32) * c->lo.n_inst : 0;
}
extern "C" hz_status hz_sha_export(hz_ctx* c, void* d_buf, void* stream) {
    if (!c || !d_buf || c->lo.sec_hi < 0) return set_err(HZ_ERR_ARG, "hz_sha_export: needs a context with a HashInputs section and a buffer");
    HZ_HIP(hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->s_main;
    const size_t mb = (size_t)c->lo.hi.sha.nblocks * 64 * c->lo.n_inst, cb = ((size_t)c->lo.hi.sha.nblocks + 1) * 32 * c->lo.n_inst;
    HZ_HIP(hipMemcpyAsync(d_buf, c->msg.p, mb, hipMemcpyDeviceToDevice, s));
    HZ_HIP(hipMemcpyAsync((uint8_t*)d_buf + mb, c->chain.p, cb, hipMemcpyDeviceToDevice, s));
    return HZ_OK;
}
extern "C" hz_status hz_sha_expand(hz_ctx* c, int32_t first, int32_t count, const void* d_buf, void* stream) {
    if (!c || c->lo.sec_hi < 0) return set_err(HZ_ERR_ARG, "hz_sha_expand: needs a context with a HashInputs section");
    const int64_t nb = c->lo.hi.sha.nblocks;
    if (first < 0 || count < 0 || (int64_t)first + count > nb) return set_err(HZ_ERR_ARG, "hz_sha_expand: blocks %d..%d out of range (%lld blocks)", first, first + count - 1, (long long)nb);
    HZ_HIP(hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->s_main;
    if (d_buf) {
        const size_t mb = (size_t)nb * 64 * c->lo.n_inst, cb = ((size_t)nb + 1) * 32 * c->lo.n_inst;
        HZ_HIP(hipMemcpyAsync(c->msg.p, d_buf, mb, hipMemcpyDeviceToDevice, s));
        HZ_HIP(hipMemcpyAsync(c->chain.p, (const uint8_t*)d_buf + mb, cb, hipMemcpyDeviceToDevice, s));
    }
    if (!c->enqueued) {   // a rank with an empty transaction shard has not reset its failure record this step
        const hz_status st = reset_err(c, s, ~0ull);
        if (st != HZ_OK) return st;
    }
    HZ_HIP(launch_sha_expand_range(make_hi(c, true), (uint32_t)first, (uint32_t)count, s));
    c->last_stream = s;
    c->enqueued = true;
    return HZ_OK;
}

extern "C" hz_status hz_ctx_set_profiling(hz_ctx* c, int32_t on) {
    if (!c) return set_err(HZ_ERR_ARG, "hz_ctx_set_profiling: null context");
    c->profiling = on != 0;
    c->exclusive = on == 2;
    return HZ_OK;
}
extern "C" int32_t hz_profile_count(const hz_ctx* c) { return c ? (int32_t)c->prof_used : 0; }
extern "C" hz_status hz_profile_get(hz_ctx* c, int32_t i, const char** kernel, float* ms, uint64_t* algorithmic_bytes, uint64_t* units) {
    if (!c || i < 0 || (size_t)i >= c->prof_used) return set_err(HZ_ERR_ARG, "hz_profile_get: bad index");
    hz_ctx::Prof& p = c->prof[i];
    HZ_HIP(hipSetDevice(c->device));
    HZ_HIP(hipEventSynchronize(p.e1));
    HZ_HIP(hipEventElapsedTime(&p.ms, p.e0, p.e1));
    if (kernel) *kernel = p.name.c_str();
    if (ms) *ms = p.ms;
    uint64_t bytes = p.bytes;
    if (c->zm_skip.p && (p.name == "smt" || p.name == "fee_smt")) {
        // what the launch left in place because the buffer held it already (constant marks) is not among the bytes it is responsible for
        unsigned long long sk[3] = {0, 0, 0};
        HZ_HIP(hipMemcpy(sk, c->zm_skip.p, sizeof sk, hipMemcpyDeviceToHost));
        const uint64_t left = (uint64_t)sk[p.name == "smt" ? 0 : 1] * 32ull;
        bytes = bytes > left ? bytes - left : 0;
    }
    if (algorithmic_bytes) *algorithmic_bytes = bytes;
    if (units) *units = p.units;
    return HZ_OK;
}

extern "C" hz_status hz_witness_run(hz_ctx* c, hz_error* err) {
    hz_status st = hz_witness_enqueue(c, nullptr);
    if (st != HZ_OK) return st;
    return hz_witness_check(c, err);
}

hz::ExportScratch* hz::ctx_export_scratch(hz_ctx* c) { return &c->exp_scratch; }
void hz::ctx_geometry(const hz_ctx* c, CtxGeom& g) {
    const Layout& lo = c->lo;
    g.nsec = (uint32_t)std::min<size_t>(lo.sections.size(), 4);
    for (uint32_t i = 0; i < g.nsec; i++) g.sec[i] = SecMap{lo.sections[i].vbase, lo.sections[i].base, lo.sections[i].upi, lo.sections[i].n_units};
    g.n_inst = lo.n_inst; g.per_instance = lo.per_instance; g.total = lo.total;
    g.device = c->device; g.s_main = c->s_main; g.wit = c->wit.p;
}

// gather `count` elements of the per-instance (virtual) witness of instance `inst` into a dense buffer
struct GatherArgs { const uint4* wit; uint4* out; uint64_t first, count; uint32_t inst, nsec; SecMap sec[4]; };
__global__ void k_gather_virtual(const GatherArgs a) {
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < a.count; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t v = a.first + t;
        uint32_t si = a.nsec - 1;
        while (si > 0 && a.sec[si].vbase > v) si--;
        const SecMap m = a.sec[si];
        const uint64_t rel = v - m.vbase;
        const uint64_t p = m.base + (rel / m.upi) * m.n_units + (uint64_t)a.inst * m.upi + rel % m.upi;
        a.out[2 * t] = a.wit[2 * p];
        a.out[2 * t + 1] = a.wit[2 * p + 1];
    }
}

extern "C" hz_status hz_witness_read(hz_ctx* c, int32_t instance, uint64_t first, uint64_t count, uint8_t* out) {
    if (!c || !out) return set_err(HZ_ERR_ARG, "hz_witness_read: null argument");
    const Layout& lo = c->lo;
    if (first > lo.per_instance || count > lo.per_instance - first) return set_err(HZ_ERR_ARG, "hz_witness_read: range beyond the witness");
    if (instance < 0 || (uint32_t)instance >= lo.n_inst) return set_err(HZ_ERR_ARG, "hz_witness_read: bad instance");
    HZ_HIP(hipSetDevice(c->device));
    if (count == 0) return HZ_OK;
    if (lo.n_inst == 1) {   // virtual == physical
        HZ_HIP(hipMemcpy(out, (const uint8_t*)c->wit.p + first * 32, count * 32, hipMemcpyDeviceToHost));
        return HZ_OK;
    }
    // chunked gather through a staging buffer
    const uint64_t chunk = std::min<uint64_t>(count, 1u << 22);
    if (c->stage.bytes < chunk * 32) HZ_HIP(c->stage.alloc(chunk * 32));
    GatherArgs g;
    memset(&g, 0, sizeof g);
    g.wit = (const uint4*)c->wit.p; g.out = (uint4*)c->stage.p; g.inst = (uint32_t)instance; g.nsec = (uint32_t)lo.sections.size();
    for (size_t i = 0; i < lo.sections.size() && i < 4; i++) g.sec[i] = SecMap{lo.sections[i].vbase, lo.sections[i].base, lo.sections[i].upi, lo.sections[i].n_units};
    for (uint64_t done = 0; done < count; done += chunk) {
        g.first = first + done;
        g.count = std::min<uint64_t>(chunk, count - done);
        hipLaunchKernelGGL(k_gather_virtual, dim3((unsigned)std::min<uint64_t>((g.count + 255) / 256, 4096)), dim3(256), 0, c->s_main, g);
        HZ_HIP(hipGetLastError());
        HZ_HIP(hipMemcpyAsync(out + done * 32, c->stage.p, g.count * 32, hipMemcpyDeviceToHost, c->s_main));
        HZ_HIP(hipStreamSynchronize(c->s_main));
    }
    return HZ_OK;
}

// arbitrary elements of the per-instance witness (the circom-ordered view of formats.hip: hz_witness_read_sym)
__global__ void k_gather_index(const GatherArgs a, const uint64_t* __restrict__ index) {
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < a.count; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t v = index[t];
        uint32_t si = a.nsec - 1;
        while (si > 0 && a.sec[si].vbase > v) si--;
        const SecMap m = a.sec[si];
        const uint64_t rel = v - m.vbase;
        const uint64_t p = m.base + (rel / m.upi) * m.n_units + (uint64_t)a.inst * m.upi + rel % m.upi;
        a.out[2 * t] = a.wit[2 * p];
        a.out[2 * t + 1] = a.wit[2 * p + 1];
    }
}
extern "C" hz_status hz_witness_gather(hz_ctx* c, int32_t instance, const uint64_t* index, uint64_t count, uint8_t* out) {
    if (!c || !index || !out) return set_err(HZ_ERR_ARG, "hz_witness_gather: null argument");
    const Layout& lo = c->lo;
    if (instance < 0 || (uint32_t)instance >= lo.n_inst) return set_err(HZ_ERR_ARG, "hz_witness_gather: bad instance");
    for (uint64_t i = 0; i < count; i++)
        if (index[i] >= lo.per_instance) return set_err(HZ_ERR_ARG, "hz_witness_gather: index %llu beyond the witness", (unsigned long long)index[i]);
    HZ_HIP(hipSetDevice(c->device));
    const uint64_t chunk = std::min<uint64_t>(std::max<uint64_t>(count, 1), 1u << 21);
    DevBuf didx, dout;
    HZ_HIP(didx.alloc(chunk * 8));
    HZ_HIP(dout.alloc(chunk * 32));
    GatherArgs g;
    memset(&g, 0, sizeof g);
    g.wit = (const uint4*)c->wit.p; g.out = (uint4*)dout.p; g.inst = (uint32_t)instance; g.nsec = (uint32_t)lo.sections.size();
    for (size_t i = 0; i < lo.sections.size() && i < 4; i++) g.sec[i] = SecMap{lo.sections[i].vbase, lo.sections[i].base, lo.sections[i].upi, lo.sections[i].n_units};
    for (uint64_t done = 0; done < count; done += chunk) {
        g.count = std::min<uint64_t>(chunk, count - done);
        HZ_HIP(hipMemcpyAsync(didx.p, index + done, g.count * 8, hipMemcpyHostToDevice, c->s_main));
        hipLaunchKernelGGL(k_gather_index, dim3((unsigned)std::min<uint64_t>((g.count + 255) / 256, 4096)), dim3(256), 0, c->s_main, g, (const uint64_t*)didx.p);
        HZ_HIP(hipGetLastError());
        HZ_HIP(hipMemcpyAsync(out + done * 32, dout.p, g.count * 32, hipMemcpyDeviceToHost, c->s_main));
        HZ_HIP(hipStreamSynchronize(c->s_main));
    }
    return HZ_OK;
}

// This library's stored signals in COMPONENT-MAJOR order, as index[] for hz_symmap_from_index: the constant, then section by section
// every unit's signals together (transaction 0's signals, transaction 1's, ...) -- the shape of a constraint-reducing circom
// compile's numbering (a component's signals are consecutive variables), without the compiler. Returns the number of entries
// (= hz_witness_len); writes min(cap, that) of them.
extern "C" uint64_t hz_component_major_index(const hz_ctx* c, uint64_t* index, uint64_t cap) {
    if (!c) return 0;
    const Layout& lo = c->lo;
    uint64_t n = 0;
    auto put = [&](uint64_t v) { if (index && n < cap) index[n] = v; n++; };
    for (size_t si = 0; si < lo.sections.size(); si++) {
        const Section& s = lo.sections[si];
        for (uint32_t u = 0; u < s.upi; u++)
            for (uint32_t sig = 0; sig < s.n_sigs; sig++) {
                const uint64_t v = lo.virt((int)si, sig, u);
                if (v == 0 && n != 0) continue;   // (the constant is variable 0 wherever the layout keeps it)
                put(v);
            }
    }
    return n;
}

// raw physical view (signal-major), for bulk consumers and tests
extern "C" hz_status hz_witness_read_raw(hz_ctx* c, uint64_t first, uint64_t count, uint8_t* out) {
    if (!c || !out) return set_err(HZ_ERR_ARG, "hz_witness_read_raw: null argument");
    if (first > c->lo.total || count > c->lo.total - first) return set_err(HZ_ERR_ARG, "hz_witness_read_raw: range beyond the buffer");
    HZ_HIP(hipSetDevice(c->device));
    if (count) HZ_HIP(hipMemcpy(out, (const uint8_t*)c->wit.p + first * 32, count * 32, hipMemcpyDeviceToHost));
    return HZ_OK;
}
extern "C" uint64_t hz_witness_total(const hz_ctx* c) { return c ? c->lo.total : 0; }

// ---- symbols ---------------------------------------------------------------------------------------------
extern "C" uint64_t hz_symbol_count(const hz_ctx* c) { return c ? c->sym_total : 0; }

extern "C" hz_status hz_symbol_get(const hz_ctx* cc, uint64_t i, hz_symbol* out) {
    hz_ctx* c = const_cast<hz_ctx*>(cc);
    if (!c || !out || i >= c->sym_total) return set_err(HZ_ERR_ARG, "hz_symbol_get: bad index");
    // binary search the block
    size_t lo_i = 0, hi_i = c->sym_index.size() - 1;
    while (lo_i < hi_i) {
        const size_t mid = (lo_i + hi_i + 1) / 2;
        if (c->sym_index[mid].first <= i) lo_i = mid;
        else hi_i = mid - 1;
    }
    const hz_ctx::BlkRef& r = c->sym_index[lo_i];
    const Section& s = c->lo.sections[r.sec];
    const Block& b = s.blocks[r.blk];
    const uint64_t rel = i - r.first;
    const uint32_t u = (uint32_t)(rel / b.count), k = (uint32_t)(rel % b.count);
    std::string nm = ssub(b.name, "{u}", istr(u));
    if (b.kind == BK_POSEIDON) nm += poseidon_signame(b.t, (int)k);
    else if (b.kind == BK_SHA) nm += "[" + istr(k / SHA_BLOCK_SIGS) + "]" + sha_signame(k % SHA_BLOCK_SIGS);
    else if (b.count > 1 || b.scalar_array) nm += "[" + istr(b.idx0 + u * b.ustride + k) + "]";   // slices keep circom's index (Block::idx0)
    c->sym_name = nm;
    out->name = c->sym_name.c_str();
    out->index = c->lo.virt(r.sec, b.off + k, u);
    return HZ_OK;
}

extern "C" int32_t hz_symbol_lookup(const hz_ctx* c, const char* name, uint64_t* index) {
    if (!c || !name) return 0;
    uint64_t idx = 0;
    if (!c->lo.lookup(name, &idx)) return 0;
    if (index) *index = idx;
    return 1;
}
