/* hermez_witness.h -- C ABI of the MI355X-native witness generator for the Hermez rollup circuits.
 *
 * This is the drop-in boundary for the one hot path this repository accelerates: witness
 * calculation for the circuits of hermeznetwork/circuits. What each entry point replaces:
 *
 *   reference call site                                         replaced by
 *   ----------------------------------------------------------  --------------------------------
 *   circuit = await tester(path)          test/rollup-main.test.js:52,  test/rollup-tx.test.js:43
 *        (circom compile + WitnessCalculator instantiate)       hz_ctx_create / hz_ctx_destroy
 *   WitnessCalculator setSignal(name, idx, value) loop inside
 *        circuit.calculateWitness(input)  test/helpers/helpers.js:142,149    hz_set_input
 *   calculateWitness body (component code of DecodeTx/RollupTx/FeeTx/HashInputs,
 *        src/rollup-main.circom:201-475)                        hz_witness_run
 *   returned witness array w[]            test/helpers/helpers.js:149        hz_witness_read
 *   circuit.assertOut / getSignal via .sym                      hz_symbol_count / hz_symbol_get /
 *                                         test/helpers/helpers.js:143,154,170    hz_symbol_lookup
 *   native witness binary `./circuit input.json witness.json`   the same five calls
 *                                         tools/helpers/actions.js:132-146
 *   Poseidon(n) gadget (circomlib 0.5.2 poseidon.circom; call sites src/lib/hash-state.circom:32,
 *        src/decode-tx.circom:275)                              hz_poseidon_batch(_dev)
 *
 * All field elements cross this boundary as 32-byte little-endian canonical integers (< r), the
 * element format of snarkjs .wtns files. Plain C types only; caller-allocated buffers; no
 * exceptions cross the ABI. A context is used by one thread at a time.
 * There is NO CPU fallback: every entry point that computes returns HZ_ERR_NODEVICE when no
 * gfx950 device is usable.
 *
 * Streams. Every `void* stream` argument is a hipStream_t. NULL never means HIP's legacy default stream: it means the
 * context's own non-blocking stream, in every entry point alike. Asynchronous input writes (hz_set_input_dev,
 * hz_copy_instance_inputs, hz_inputs_upload) are ordered before the next hz_witness_enqueue / hz_witness_run of the same
 * context on whatever stream that call uses (an event, no host synchronisation). Everything else follows stream order:
 * hz_da_export / hz_da_import / hz_witness_enqueue_tail must be given the stream of the hz_witness_enqueue they follow, and a
 * collective between them must be ordered on that stream by the caller.
 */
#ifndef HERMEZ_WITNESS_H
#define HERMEZ_WITNESS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define HZ_FR_BYTES 32

typedef enum {
    HZ_OK = 0,
    HZ_ERR_ARG = 1,        /* bad parameter */
    HZ_ERR_HIP = 2,        /* HIP runtime failure, see hz_last_error() */
    HZ_ERR_CONSTRAINT = 3, /* a `===` of the circuit does not hold; details in hz_error */
    HZ_ERR_INPUT = 4,      /* unknown / mis-shaped / missing input signal */
    HZ_ERR_NODEVICE = 5    /* no usable gfx950 device */
} hz_status;

/* main component selector: the templates the reference's suites instantiate as `component main` */
typedef enum {
    HZ_T_ROLLUP_MAIN = 0, /* RollupMain(nTx,nLevels,maxL1Tx,maxFeeTx)  src/rollup-main.circom:82   */
    HZ_T_ROLLUP_TX = 1,   /* RollupTx(nLevels,maxFeeTx)                src/rollup-tx.circom:78     */
    HZ_T_DECODE_TX = 2,   /* DecodeTx(nLevels)                         src/decode-tx.circom:44     */
    HZ_T_FEE_TX = 3,      /* FeeTx(nLevels)                            src/fee-tx.circom:26        */
    HZ_T_HASH_STATE = 4,  /* HashState()                               src/lib/hash-state.circom:18*/
    HZ_T_WITHDRAW = 5,    /* Withdraw(nLevels)                         src/withdraw.circom:21      */
    HZ_T_HASH_INPUTS = 6, /* HashInputs(nLevels,nTx,maxL1Tx,maxFeeTx)  src/hash-inputs.circom:23   */
    /* gadget mains of the reference's unit suites (one witness per instance, a single small kernel) */
    HZ_T_DECODE_FLOAT = 7,     /* DecodeFloat()                        src/lib/decode-float.circom:51   */
    HZ_T_COMPUTE_FEE = 8,      /* ComputeFee()                         src/compute-fee.circom:12        */
    HZ_T_FEE_ACCUMULATOR = 9,  /* FeeAccumulator(maxFeeTx)             src/fee-accumulator.circom:56    */
    HZ_T_BALANCE_UPDATER = 10, /* BalanceUpdater()                     src/balance-updater.circom:24    */
    HZ_T_ROLLUP_TX_STATES = 11,/* RollupTxStates()                     src/rollup-tx-states.circom:39   */
    HZ_T_RQ_TX_VERIFIER = 12,  /* RqTxVerifier()                       src/rq-tx-verifier.circom:19     */
    HZ_T_MUX256 = 13,          /* Mux256()                             src/lib/mux256.circom:10         */
    HZ_T_BITS2AYSIGN = 14,     /* BitsCompressed2AySign()              src/lib/utils-bjj.circom:12      */
    HZ_T_AYSIGN2AX = 15,       /* AySign2Ax()                          src/lib/utils-bjj.circom:37      */
    /* circomlib 0.5.2's own sparse-Merkle-tree templates as main: the components behind src/rollup-tx.circom:537-570,
     * src/fee-tx.circom:97 (SMTProcessor(nLevels+1)) and src/withdraw.circom:47-58 (SMTVerifier(nLevels+1)). Here nLevels is the
     * template's own parameter (number of levels). Inputs as in circomlib: oldRoot, siblings[n], oldKey, oldValue, isOld0, newKey,
     * newValue, fnc[2] -> newRoot; enabled, root, siblings[n], oldKey, oldValue, isOld0, key, value, fnc. */
    HZ_T_SMT_PROCESSOR = 16,
    HZ_T_SMT_VERIFIER = 17,
    HZ_T_COUNT
} hz_template;

typedef struct {
    int32_t template_id; /* hz_template */
    int32_t nTx;         /* RollupMain / HashInputs only */
    int32_t nLevels;
    int32_t maxL1Tx;
    int32_t maxFeeTx;
    int32_t device;      /* HIP device ordinal */
    /* Number of independent instances of the main component evaluated by one hz_witness_run
     * (e.g. 2^20 Withdraw witnesses, or several RollupMain batches in flight). Inputs and the
     * witness carry the instance as the outermost index. 0 means 1. */
    int32_t n_instances;
    int32_t flags;       /* 0, or one of HZ_FLAG_* */
} hz_params;
/* HZ_FLAG_LATENCY: a RollupMain context of one to four batches puts its concurrent kernel chains -- front/hash/SMT/HashInputs, the
 * two signature kernels, the fee chain -- on disjoint sets of compute units through CU-masked streams: one 2048-transaction batch
 * alone 14.4 -> 7.7 ms (7.3 with HZ_FLAG_SOLO). Up to four such contexts in flight overlap (every context uses the same four masks): one batch each, 450 k
 * tx-witnesses/s with two, 640 k with four; plain contexts do not overlap in that regime (250 k). While at most TWO such contexts of one
 * batch are alive on the device their SMT chain kernel takes the latency form HZ_FLAG_SOLO describes (decided per launch; with four
 * in flight it would cost 15 %: profiles/r06_flagged_latency_form.txt). From eight batches per launch on,
 * the default (0 or HZ_FLAG_THROUGHPUT: any kernel on any CU) is faster. Every such context holds four hardware queues whose scratch
 * memory the runtime sizes by their hungriest kernel: within one process FOUR contexts per device get the partition by default
 * (HZ_MAX_PARTITIONED=<n> in the environment changes it; two until round 6, when a front kernel with 7.7 KB of scratch per lane made
 * four of them abort in the ROCm 7 runtime: profiles/r05_latency_regime.txt), further ones silently get the default schedule
 * (same witness). */
#define HZ_FLAG_THROUGHPUT 1
#define HZ_FLAG_LATENCY 2
/* HZ_FLAG_SOLO (with HZ_FLAG_LATENCY, contexts of ONE batch: no effect on larger ones): nothing else runs on the device while this context's step
 * does -- no second context in flight. Its SMT chain kernel then takes the latency form (a quad of lanes per chain, the level hash
 * spread over the quad: 0.67 x the chain's time for 2.5 x its instructions and a whole CU partition's wavefront slots -- which is why
 * it is wrong beside MANY other contexts: one batch x 4 contexts 540 k tx/s with it, 640 k without; a HZ_FLAG_LATENCY context takes it by
 * itself while at most two are alive; HZ_FLAG_SOLO keeps it whatever else exists). */
#define HZ_FLAG_SOLO 4

typedef struct {
    int32_t instance;      /* which instance failed */
    int32_t unit;          /* transaction / fee-tx index inside the instance, -1 if global */
    int32_t constraint_id; /* stable id, see hz_constraint_name() */
    uint8_t lhs[HZ_FR_BYTES];
    uint8_t rhs[HZ_FR_BYTES];
} hz_error;

typedef struct {
    const char* name; /* circom-style dotted name, e.g. main.rollupTx[3].processor1.newRoot */
    uint64_t index;   /* position in the flat witness of instance 0 */
} hz_symbol;

typedef struct hz_ctx hz_ctx;

/* library / device ----------------------------------------------------------------------- */
const char* hz_version(void);
const char* hz_last_error(void);          /* thread-local text of the last failure */
int32_t hz_device_count(void);            /* 0 when no gfx950 device is usable */

/* context: one compiled "circuit" ---------------------------------------------------------- */
hz_status hz_ctx_create(const hz_params* params, hz_ctx** out);
void hz_ctx_destroy(hz_ctx* ctx);
/* number of field elements in the witness of ONE instance (w[0] == 1 included) */
uint64_t hz_witness_len(const hz_ctx* ctx);
/* bytes of device memory the context holds, counting the buffers that are only allocated on first use (upload staging, the
 * signature ladder's side buffer of small launches): what n_instances of this template cost of a device's 288 GB */
uint64_t hz_ctx_device_bytes(const hz_ctx* ctx);
/* the same figure for a context that does not exist yet (no device needed): hz_template_device_bytes(&params) */
uint64_t hz_template_device_bytes(const hz_params* params);
/* closed-form constraint count of the template (reference tools/circuit-constraints.js:31-75) */
uint64_t hz_constraint_estimate(const hz_ctx* ctx);

/* inputs: `name` is a main-component input signal (e.g. "siblings1"); `vals` holds `count`
 * canonical 32-byte LE integers in circom's row-major flattening of the signal's dimensions,
 * for instance `instance`. For templates evaluated over n_instances, `instance` = -1 sets all
 * instances at once (`vals` = [n_instances][flat_len]). Values >= r are rejected. */
hz_status hz_set_input(hz_ctx* ctx, int32_t instance, const char* name, const uint8_t* vals, size_t count);
/* same, but `vals` already lives in device memory of ctx's device */
hz_status hz_set_input_dev(hz_ctx* ctx, int32_t instance, const char* name, const void* dvals, size_t count, void* stream);
/* forget which inputs were set (values are kept; calculateWitness semantics need a fresh set) */
/* Replicate all input signals of instance `src` onto instance `dst`, device to device, on `stream`.
 * (No reference counterpart: snarkjs computes one witness per call; this fills a multi-instance context.) */
hz_status hz_copy_instance_inputs(hz_ctx* ctx, int32_t src, int32_t dst, void* stream);
/* Forget which inputs were set (the next enqueue wants every one again) and whatever the library remembers of the buffer's content:
 * the step after it stores every signal, including the constants an earlier step left in place (DESIGN.md "constant marks"). */
void hz_clear_inputs(hz_ctx* ctx);
/* Bulk input path (the marshalling half of calculateWitness(input), reference test/helpers/helpers.js:147-149, for a process that
 * feeds batch after batch): ONE packed buffer per instance instead of one call per signal. Layout: every input signal in
 * hz_input_name order at hz_input_packed_offset(i) (32-byte aligned), as [outer][inner] little-endian elements -- the flattening
 * hz_set_input takes -- of hz_input_packed_width(i) bytes each: 32, or 1 for bit-valued signals (fromBjjCompressed).
 * hz_inputs_upload copies the buffer to the device asynchronously on `stream` (truly so from hz_host_alloc'ed pinned memory, which
 * the caller must keep unchanged until the stream reaches the copy) and one kernel scatters it into the witness layout and
 * range-checks the 32-byte elements; an element >= r is reported by the next hz_witness_check as HZ_ERR_INPUT.
 * `packed` may also be DEVICE memory (hz_inputs_upload / _stage / _stage_range alike): a host that keeps the packed inputs of its
 * batches resident in HBM hands each step's over with a device-to-device copy (bench.py's `value` loop). */
uint64_t hz_inputs_packed_bytes(const hz_ctx* ctx);
int32_t hz_input_packed_width(const hz_ctx* ctx, int32_t i);
uint64_t hz_input_packed_offset(const hz_ctx* ctx, int32_t i);
void* hz_host_alloc(size_t bytes);
void hz_host_free(void* p);
hz_status hz_inputs_upload(hz_ctx* ctx, int32_t instance, const void* packed, size_t bytes, void* stream);
/* The copy alone, ahead of time: the packed buffer goes to the instance's device staging slot on `stream` (NULL = the context's
 * copy stream) and the NEXT hz_witness_enqueue / hz_witness_run scatters every staged instance into the witness layout before
 * its kernels. Called right after the enqueue of step N with the inputs of step N + 1, the PCIe transfer overlaps step N's
 * kernels, which still read the old inputs. All stage calls between two enqueues of a context must use the same stream. */
hz_status hz_inputs_stage(hz_ctx* ctx, int32_t instance, const void* packed, size_t bytes, void* stream);
/* The same for `count` consecutive instances, instance first + j from packed + j * stride. When the host buffers are contiguous
 * (stride == bytes_each) the whole range crosses PCIe as one copy: what a serving loop calls once per step. */
hz_status hz_inputs_stage_range(hz_ctx* ctx, int32_t first, int32_t count, const void* packed, size_t bytes_each, size_t stride, void* stream);
/* enumerate the input signals the template expects */
int32_t hz_input_count(const hz_ctx* ctx);
const char* hz_input_name(const hz_ctx* ctx, int32_t i, uint64_t* flat_len);

/* run: all instances, inputs resident in HBM. Asynchronous variant enqueues on `stream` and
 * returns; hz_witness_check then synchronises and reports the first violated constraint
 * (lowest instance, lowest unit, lowest constraint id). hz_witness_run = enqueue + check. */
hz_status hz_witness_enqueue(hz_ctx* ctx, void* stream);
hz_status hz_witness_check(hz_ctx* ctx, hz_error* err);
hz_status hz_witness_run(hz_ctx* ctx, hz_error* err);
/* After hz_witness_check / hz_witness_run: the first violated constraint of EVERY instance of that launch, ordered by instance
 * (out[0] is what hz_witness_check reported). The reference evaluates one circuit per calculateWitness call and throws at its first
 * failing `===` (test/rollup-main.test.js:868-877, test/rollup-tx.test.js:911-918); a launch here evaluates n_instances of them, and
 * a serving loop has to know every batch to reject, not only the first. *n_failed = number of failing instances (0 when the launch
 * was clean), of which min(cap, *n_failed) are written. cap = 0 only counts; otherwise the operands cost one more pass over the
 * kernels when there are failures. */
hz_status hz_witness_failures(hz_ctx* ctx, hz_error* out, size_t cap, size_t* n_failed);

/* Per-kernel timing of the last enqueue, measured with HIP events on the launch stream.
 * `algorithmic_bytes` = 32 B x (witness signals the kernel is responsible for) x units.
 * on = 1: the normal schedule (EdDSA and fee chains overlap the hash/SMT chain on side streams);
 * on = 2: exclusive, every kernel alone on the device (durations usable for a per-kernel roofline). */
hz_status hz_ctx_set_profiling(hz_ctx* ctx, int32_t on);
int32_t hz_profile_count(const hz_ctx* ctx);
hz_status hz_profile_get(hz_ctx* ctx, int32_t i, const char** kernel, float* ms, uint64_t* algorithmic_bytes, uint64_t* units);

/* output: copy `count` elements starting at flat index `first` of instance `instance` to host */
hz_status hz_witness_read(hz_ctx* ctx, int32_t instance, uint64_t first, uint64_t count, uint8_t* out);
/* The physical buffer: sections stored signal-major (include/hz_layout.h); `total` elements. For
 * instanced templates element (signal s, instance k) sits at s * n_instances + k. */
uint64_t hz_witness_total(const hz_ctx* ctx);
hz_status hz_witness_read_raw(hz_ctx* ctx, uint64_t first, uint64_t count, uint8_t* out);
const void* hz_witness_dev_ptr(const hz_ctx* ctx);

/* wire / disk formats ------------------------------------------------------------
 * hz_set_inputs_json: the input.json the reference's tools write (tools/generate-input.js:109; decimal strings, bare
 *   integers, 0x-hex, nested arrays; values reduced mod r) -> hz_set_input per key.
 * hz_witness_write_json: witness.json, array of decimal strings (tools/helpers/actions.js:136-139).
 * hz_witness_write_wtns: snarkjs .wtns (version 2: header section {n8 = 32, prime, nVars}, data section nVars x 32 B LE).
 * hz_symbols_write_sym: circom .sym lines `labelIdx,varIdx,componentIdx,name` for the stored signals. */
hz_status hz_set_inputs_json(hz_ctx* ctx, int32_t instance, const char* json, size_t len);
hz_status hz_witness_write_json(hz_ctx* ctx, int32_t instance, const char* path);
hz_status hz_witness_write_wtns(hz_ctx* ctx, int32_t instance, const char* path);
hz_status hz_symbols_write_sym(const hz_ctx* ctx, const char* path);

/* circom .sym import: the witness in the COMPILER's variable order. The reference's prover steps (snarkjs / rapidsnark on the
 * r1cs + zkey of tools/helpers/actions.js:30-68,148-170) consume circom's numbering, which only the compiler's .sym records
 * (`labelIdx,varIdx,componentIdx,name`; varIdx = -1 for eliminated signals; wired labels share a varIdx). hz_symmap_create joins
 * it by name with the signals this library stores: a variable resolves when any of its labels is stored. hz_symmap_unresolved
 * returns how many variables did not (and the i-th one's number and a label): a compile that keeps signals this layout drops
 * cannot be served. hz_witness_read_sym / hz_witness_write_wtns_sym then deliver the witness in that order;
 * hz_witness_gather reads arbitrary positions of this library's own per-instance numbering. */
typedef struct hz_symmap hz_symmap;
hz_status hz_symmap_create(const hz_ctx* ctx, const char* sym_text, size_t len, hz_symmap** out);
void hz_symmap_destroy(hz_symmap* map);
uint64_t hz_symmap_nvars(const hz_symmap* map);
uint64_t hz_symmap_unresolved(const hz_symmap* map, uint64_t i, uint64_t* var, const char** name);
/* how many variables of the map are DERIVED: linear signals this layout does not store and a compile without constraint reduction
 * keeps (reference test/rollup-main.test.js:52, reduceConstraints:false) -- every signal inside a Poseidon component (ark / mix /
 * S-box inputs, from the stored S-box products), and the linear intermediates / linearly fed component inputs of the reference's own
 * templates (rule table in csrc/formats.hip) -- evaluated from stored signals when the witness is read in the compiler's order */
uint64_t hz_symmap_derived(const hz_symmap* map);
/* .sym AND .r1cs of the same compile (tools/helpers/actions.js:30-68 writes both). Variables that no label resolves -- the
 * wire-through signals an unreduced compile keeps: inputs of sub-components, aliases of their outputs, Bits2Num sums, comparator
 * differences, for ANY circuit and circomlib version -- are solved from the circuit's own LINEAR constraints: one with exactly one
 * unknown variable defines it over known ones, and so on until nothing changes (hz_symmap_solved = how many); a product constraint
 * A * B = C whose A and B are known and whose C holds one unknown variable defines that one too (a product signal the layout does
 * not store under that name, e.g. MultiMux4's terms over constant inputs). The r1cs stays with
 * the map: hz_symmap_check_r1cs evaluates every constraint (A.w)(B.w) = C.w on the witness as the map serves it -- what
 * `snarkjs wtns check` would do with the .wtns this library writes -- and returns the number of violated constraints and the
 * indices of the first `cap` of them. r1cs: iden3 binary format version 1, field BN254 Fr; circom's wire w is variable w. */
hz_status hz_symmap_create_r1cs(const hz_ctx* ctx, const char* sym_text, size_t sym_len, const uint8_t* r1cs, size_t r1cs_len, hz_symmap** out);
uint64_t hz_symmap_solved(const hz_symmap* map);
hz_status hz_symmap_check_r1cs(hz_ctx* ctx, const hz_symmap* map, int32_t instance, uint64_t* n_bad, uint64_t* first_bad, uint64_t cap);
/* A resolved map on disk: importing the files of a full-size circuit takes minutes, the map is a few arrays. A map belongs to one
 * template and shape (checked on load); the constraint system is not kept (hz_symmap_check_r1cs needs a map made from the files). */
hz_status hz_symmap_save(const hz_ctx* ctx, const hz_symmap* map, const char* path);
hz_status hz_symmap_load(const hz_ctx* ctx, const char* path, hz_symmap** out);
hz_status hz_witness_read_sym(hz_ctx* ctx, const hz_symmap* map, int32_t instance, uint64_t first_var, uint64_t count, uint8_t* out);
hz_status hz_witness_write_wtns_sym(hz_ctx* ctx, const hz_symmap* map, int32_t instance, const char* path);
hz_status hz_witness_gather(hz_ctx* ctx, int32_t instance, const uint64_t* index, uint64_t count, uint8_t* out);

/* The witness in the compiler's order ON THE DEVICE (SURVEY 8a' K8): what a prover on the same GPU consumes. The reference hands
 * w[] over in circom's numbering (test/helpers/helpers.js:142,149) and its prove step reads that vector beside the .r1cs / zkey
 * (tools/helpers/actions.js:132-170); the kernels of a step write a signal-major layout of their own (include/hz_layout.h).
 *   hz_symmap_upload            takes the map to the context's device once (the first export does it implicitly): per 128-byte line of
 *                               the physical buffer -- four consecutive units of one signal -- the four variables it feeds, ordered
 *                               so that consecutive lanes write consecutive variables; CSR tables of the derived variables.
 *                               device_bytes (optional) = size of those tables.
 *   hz_witness_export_dev       ONE pass: variable v of instance `instance` -> d_out[v] (hz_symmap_nvars x 32 bytes, canonical LE),
 *                               stored variables gathered, derived ones (an unreduced compile's linear signals) evaluated on the
 *                               device. instance = -1: every instance, d_out[instance][v]. Asynchronous on `stream` (NULL = the
 *                               context's stream), ordered after a hz_witness_enqueue on the same stream.
 *   hz_witness_export_range_dev the same for instances [first, first + count)
 *   hz_witness_export_host      the same pass, then asynchronous copies through a ring of two pinned buffers into `out`
 *                               (count x 32 bytes from variable first_var on); PCIe bound: 3.86 GB per headline batch at <= 63 GB/s.
 *                               hz_witness_write_wtns_sym and hz_symmap_check_r1cs run on exactly this buffer.
 *   hz_symmap_dev_index         for a consumer that prefers indirection over a copy: phys0[v] = element of hz_witness_dev_ptr() that
 *                               holds variable v of instance 0, inst_stride[v] = elements to add per instance; a derived variable
 *                               has phys0[v] = 2^63 | k, its value is element (instance slot) * n_derived_slots + k of the buffer
 *                               hz_witness_derive_dev fills (instance = -1: slot = instance; otherwise slot 0). */
/* A map given explicitly -- variable v is signal index[v] of this library's per-instance numbering, variable 0 the constant --
 * and the stored signals in component-major order (every unit's signals together: the shape of a reducing compile's numbering)
 * as such an index: hz_component_major_index returns the number of entries and writes min(cap, that). */
hz_status hz_symmap_from_index(const hz_ctx* ctx, const uint64_t* index, uint64_t n, hz_symmap** out);
uint64_t hz_component_major_index(const hz_ctx* ctx, uint64_t* index, uint64_t cap);
hz_status hz_symmap_upload(hz_ctx* ctx, const hz_symmap* map, uint64_t* device_bytes);
hz_status hz_witness_export_dev(hz_ctx* ctx, const hz_symmap* map, int32_t instance, void* d_out, void* stream);
hz_status hz_witness_export_range_dev(hz_ctx* ctx, const hz_symmap* map, int32_t first_instance, int32_t count, void* d_out, void* stream);
hz_status hz_witness_export_host(hz_ctx* ctx, const hz_symmap* map, int32_t instance, uint64_t first_var, uint64_t count, uint8_t* out);
hz_status hz_symmap_dev_index(hz_ctx* ctx, const hz_symmap* map, const uint64_t** d_phys0, const uint32_t** d_inst_stride, uint64_t* n_derived_slots);
hz_status hz_witness_derive_dev(hz_ctx* ctx, const hz_symmap* map, int32_t instance, const void** d_derived, void* stream);

/* symbols ------------------------------------------------------------------------------------ */
uint64_t hz_symbol_count(const hz_ctx* ctx);
hz_status hz_symbol_get(const hz_ctx* ctx, uint64_t i, hz_symbol* out);
/* returns 1 and the index if `name` (with or without the "main." prefix) is a witness signal */
int32_t hz_symbol_lookup(const hz_ctx* ctx, const char* name, uint64_t* index);
const char* hz_constraint_name(int32_t constraint_id);

/* Evaluates a DAG of Poseidon hashes on the device, level by level: the Merkle work of a batch builder (the counterpart of
 * @hermeznetwork/commonjs RollupDB / BatchBuilder, called by the reference at test/helpers/helpers.js:46,148 and
 * tools/generate-input.js:70-107, which hashes 2*(nLevels+1) dependent nodes per transaction on one CPU thread).
 * `vals` is a table of n_vals field elements (32-byte little-endian, canonical) in host memory: known values on entry, every
 * job's digest on return. Job j hashes the seg_t-1 elements vals[job_in[6*j + k]] and stores Poseidon(seg_t) at vals[job_out[j]].
 * Jobs are listed in execution order in segments [seg_first, seg_first+seg_count) of one width seg_t (2..7); the jobs of a segment
 * must not depend on each other (all node versions of one tree level form one segment). device_ms (optional) receives the
 * device time of the segment launches. */
#define HZ_DAG_MAX_IN 6
hz_status hz_poseidon_dag(int32_t device, uint8_t* vals, uint64_t n_vals, const uint32_t* job_in, const uint32_t* job_out, uint64_t n_jobs,
                          const uint32_t* seg_t, const uint64_t* seg_first, const uint64_t* seg_count, uint32_t n_segs, double* device_ms);

/* Poseidon batch: n independent permutations of width t = n_inputs + 1 (2..7). ----------------
 * `in`  : [n][t-1] canonical elements; `out`: [n] digests (state[0] after the last round).
 * If `sbox_witness` is non-NULL it receives the S-box signals (in2,in4,out per S-box, the
 * non-linear signals of circomlib's Poseidon template) as [3*(8t+R_P)][n] elements. */
hz_status hz_poseidon_batch(int32_t device, int32_t t, size_t n, const uint8_t* in, uint8_t* out, uint8_t* sbox_witness);
hz_status hz_poseidon_batch_dev(int32_t t, size_t n, const void* d_in, void* d_out, void* d_sbox_witness, void* stream);

/* Field self test (SURVEY 8a' K0): out[i] = a[i] (op) b[i] over BN254 Fr, one operation per lane, canonical operands and results.
 * No reference counterpart (the reference's field is ffiasm's Fr, tools/helpers/actions.js:207-215); used by tests/ only. */
enum { HZ_FR_ADD = 0, HZ_FR_SUB = 1, HZ_FR_MUL = 2, HZ_FR_SQR = 3, HZ_FR_INV = 4 /* inverse(0) = 0 */, HZ_FR_MULADD = 5 /* a*b + a + b */, HZ_FR_MIX = 6 /* 2a * (-b) */,
       HZ_FR_SQRT = 7 /* circomlib pointbits.circom sqrt(): the root <= (r-1)/2 of a, 0 when a is a non-residue (AySign2Ax, src/lib/utils-bjj.circom:37-58) */ };
hz_status hz_fr_ops(int32_t device, int32_t op, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out);

/* multi-GPU, one process per GPU (reference src/rollup-main.circom:93-99: every DecodeTx / RollupTx is
 * independent given the im* inputs). A RollupMain batch is sharded by transaction index:
 *   hz_shard_range      contiguous range of a rank
 *   hz_ctx_set_shard    this context evaluates only [first, first+count); tail != 0 on the rank that
 *                       also evaluates the fee transactions and HashInputs. count = 0 is a legal (empty) shard -- more
 *                       ranks than transactions --, count < 0 returns the context to the whole batch
 *   hz_da_export        pack the shard's per-transaction data-availability records (hz_da_record_bytes
 *                       each: L1TxFullData / L1L2TxData bits, outIdx, newExitRoot) into a device buffer --
 *                       the only data HashInputs needs from other ranks (one RCCL all_gather, ~47-330 KB)
 *   hz_da_import        write received records into this context's witness
 *   hz_witness_enqueue_tail   FeeTx + HashInputs after the imports; then hz_witness_check */
void hz_shard_range(int32_t nTx, int32_t world, int32_t rank, int32_t* first, int32_t* count);
hz_status hz_ctx_set_shard(hz_ctx* ctx, int32_t first, int32_t count, int32_t tail);
uint64_t hz_da_record_bytes(const hz_ctx* ctx);
hz_status hz_da_export(hz_ctx* ctx, void* d_buf, void* stream);
hz_status hz_da_import(hz_ctx* ctx, int32_t first, int32_t count, const void* d_buf, void* stream);
hz_status hz_witness_enqueue_tail(hz_ctx* ctx, void* stream);
/* The tail split over the ranks (SURVEY 8e "or scatter blocks back: 766 / 8"): the SHA-256 chain of HashInputs is sequential (rank 0),
 * the bit-level witness of its blocks -- 0.73 GB of the 0.76 GB the tail writes at (2048, 32) -- is independent per block given the
 * message block and the chaining value that enters it:
 *   hz_witness_enqueue_tail_chain   rank 0, after the imports: FeeTx, the message, the chain, the public output; no block witness
 *   hz_sha_blocks / hz_sha_state_bytes   number of blocks; bytes of (message blocks, chaining values): 96 B per block + 32
 *   hz_sha_export                   rank 0: that state into a device buffer (ONE broadcast, 73 KB at 766 blocks)
 *   hz_sha_expand                   any rank: take the state from the buffer (NULL on the rank that computed it) and write the
 *                                   witness of blocks [first, first + count) into this context's HashInputs section
 * Block ranges: hz_shard_range(hz_sha_blocks(ctx), world, rank, ...). The witness stays sharded: rank r holds its blocks' signals. */
hz_status hz_witness_enqueue_tail_chain(hz_ctx* ctx, void* stream);
uint64_t hz_sha_blocks(const hz_ctx* ctx);
uint64_t hz_sha_state_bytes(const hz_ctx* ctx);
hz_status hz_sha_export(hz_ctx* ctx, void* d_buf, void* stream);
hz_status hz_sha_expand(hz_ctx* ctx, int32_t first, int32_t count, const void* d_buf, void* stream);

/* The whole sharded pass inside the library, for a host in any language (the N-API addon binds it: circuit.shardStep). Counterpart of
 * the `-n` thread-per-component mode of the reference's compiled witness calculator (tools/helpers/actions.js:39-45).
 *   hz_comm_create   one per rank. transport HZ_COMM_RCCL: librccl.so is loaded with dlopen (no link-time dependency; HZ_RCCL_LIB names
 *                    another file), rank 0's ncclGetUniqueId travels over the rendezvous, ncclCommInitRank on `device`; the collectives
 *                    are ncclAllGather / ncclBroadcast on the pass's stream (xGMI). HZ_COMM_SOCKET: the two collectives staged through
 *                    host memory over the rendezvous itself (47-330 KB + 73 KB per pass: latency-sized) -- for boxes without RCCL and for
 *                    tests. rendezvous_path: a Unix socket name every rank of the node can reach (rank 0 listens on it, the others
 *                    connect, HZ_COMM_TIMEOUT_S seconds of patience, default 120); may be NULL when world == 1.
 *   hz_shard_step    one pass on `stream` (NULL: the context's own): this rank's transactions, all_gather of the data-availability records, rank 0's
 *                    imports + FeeTx + message + SHA-256 chain, broadcast of the block states, this rank's share of the block
 *                    witness. The first call shards the context (hz_ctx_set_shard with hz_shard_range(nTx, world, rank)) and
 *                    allocates the exchange buffers; every rank calls hz_witness_check afterwards. A communicator serves one context. */
typedef struct hz_comm hz_comm;
enum { HZ_COMM_RCCL = 1, HZ_COMM_SOCKET = 2 };
hz_status hz_comm_create(int32_t transport, int32_t device, int32_t rank, int32_t world, const char* rendezvous_path, hz_comm** out);
void hz_comm_destroy(hz_comm* comm);
int32_t hz_comm_rank(const hz_comm* comm);
int32_t hz_comm_world(const hz_comm* comm);
hz_status hz_shard_step(hz_ctx* ctx, hz_comm* comm, void* stream);
int32_t hz_ctx_ntx(const hz_ctx* ctx);   /* transactions per batch of a RollupMain context (0 otherwise) */

#ifdef __cplusplus
}
#endif
#endif
