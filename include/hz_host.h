/* libhz_host.so -- the HOST side of the batch builder: what a rollup coordinator runs to turn transactions into circuit inputs.
 * Counterpart of @hermeznetwork/commonjs RollupDB / BatchBuilder (not on disk), which the reference calls at
 *   test/helpers/helpers.js:46,148      rollupDb.buildBatch(...), bb.addTx(tx), bb.build(), bb.getInput(), bb.getHashInputs()
 *   tools/generate-input.js:70-107      the same calls for the synthetic benchmark batch
 * Caller-side code: it never computes a witness and uses nothing from oracle/. It has no HIP dependency of its own: the Merkle
 * hashing of a batch is recorded as a DAG and handed to an evaluator -- hz_poseidon_dag of libhermez_witness.so (one device launch
 * per tree level) when the caller installs it with hzb_db_set_dag, the library's own host Poseidon otherwise (tests, tools).
 * All field elements are 32-byte little-endian canonical integers. Not thread-safe per database. */
#ifndef HZ_HOST_H
#define HZ_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- arithmetic (used by circuits_amd/builder.py and the self tests) ------------------------------------------------------------ */
int hzb_poseidon(int n_in, const uint8_t* in, uint8_t* out);                             /* Poseidon(n_in + 1), n_in = 1..6 */
int hzb_poseidon_many(int n_in, uint64_t count, const uint8_t* in, uint8_t* out);        /* count independent hashes */
int hzb_bjj_mul(const uint8_t* px, const uint8_t* py, const uint8_t* k, uint8_t* ox, uint8_t* oy);   /* BabyJubjub k * P, affine */
int hzb_bjj_add(const uint8_t* px, const uint8_t* py, const uint8_t* qx, const uint8_t* qy, uint8_t* ox, uint8_t* oy);
int hzb_bjj_mul_base8(const uint8_t* k, uint8_t* ox, uint8_t* oy);                         /* k * Base8: 8-bit fixed-base windows */
/* count scalars -> count affine points; threads = 0: up to 8 host threads (HZB_THREADS overrides), one inversion for all */
int hzb_bjj_mul_base8_many(uint64_t count, const uint8_t* k, uint8_t* ox, uint8_t* oy, int32_t threads);
int hzb_poseidon_dev9(int n_in, const uint8_t* in, uint8_t* out);                        /* self test: the device-form permutation on the host */
int hzb_fr_inv(const uint8_t* x, uint8_t* safegcd_out, uint8_t* fermat_out);             /* self test */
/* EdDSA-Poseidon signature of `msg` under the private scalar `key` (deterministic nonce: SHA-512(key || msg) mod l, as
 * circuits_amd/builder.py Account.sign_msg): S * B8 == R8 + 8 * H(R8, A, msg) * A (circomlib eddsaposeidon.circom) */
int hzb_eddsa_sign(const uint8_t* key, const uint8_t* msg, uint8_t* r8x, uint8_t* r8y, uint8_t* s);
int hzb_eddsa_pubkey(const uint8_t* key, uint8_t* ax, uint8_t* ay);

/* ---- batch builder ------------------------------------------------------------------------------------------------------------ */
typedef struct hzb_db hzb_db;
typedef struct hzb_batch hzb_batch;
const char* hzb_last_error(void);
enum { HZB_OK = 0, HZB_ERR_ARG = 1, HZB_ERR_REJECTED = 2 /* a transaction the circuit would reject */, HZB_ERR_EVAL = 3 };

/* the signature of hz_poseidon_dag (include/hermez_witness.h) */
typedef int (*hzb_dag_fn)(int32_t device, uint8_t* vals, uint64_t n_vals, const uint32_t* job_in, const uint32_t* job_out, uint64_t n_jobs,
                          const uint32_t* seg_t, const uint64_t* seg_first, const uint64_t* seg_count, uint32_t n_seg, double* device_ms);

typedef struct hzb_leaf {   /* one account (reference src/lib/hash-state.circom:14-40) */
    uint32_t token_id, sign;
    uint64_t nonce;
    uint8_t balance[32], ay[32], eth_addr[32];
} hzb_leaf;

enum { HZB_TX_HAS_AUX_TO = 1, HZB_TX_HAS_NONCE = 2, HZB_TX_HAS_RQ = 4, HZB_TX_HAS_SIG = 8, HZB_TX_HAS_SIGNER = 16 };
typedef struct hzb_tx {   /* the fields of the reference's tx objects (test/rollup-main.test.js, tools/helpers/gen-inputs-utils.js) */
    uint64_t from_idx, to_idx, aux_to_idx /* HAS_AUX_TO; else looked up for transfers to an address */;
    uint64_t amount_f, load_amount_f /* float40 */, nonce /* HAS_NONCE; else the sender's next */;
    uint32_t token_id, max_num_batch;
    uint8_t on_chain, user_fee, rq_offset, to_bjj_sign;
    uint32_t flags;
    uint8_t to_eth_addr[32], to_bjj_ay[32], from_eth_addr[32], from_bjj_compressed[32];
    uint8_t rq_tx_compressed_data_v2[32], rq_to_eth_addr[32], rq_to_bjj_ay[32];   /* HAS_RQ; else derived from rq_offset */
    uint8_t r8x[32], r8y[32], s[32];                                               /* HAS_SIG */
    uint8_t signer_key[32];                                                        /* HAS_SIGNER: the builder signs (synthetic batches) */
} hzb_tx;

hzb_db* hzb_db_create(uint32_t chain_id, uint64_t first_idx);
/* a working copy (the reference's suites build a batch on a copy of the state and consolidate it afterwards). hzb_batch_build updates
 * its database in place while it walks the batch: when it fails (HZB_ERR_REJECTED: a transaction the circuit would reject; HZB_ERR_EVAL;
 * memory) the database is left half-updated and every later call on it returns HZB_ERR_REJECTED -- build on a clone when a batch may
 * be rejected, and keep the original. */
hzb_db* hzb_db_clone(const hzb_db* db);
void hzb_db_destroy(hzb_db* db);
/* fn = hz_poseidon_dag (or NULL: host hashing) */
int hzb_db_set_dag(hzb_db* db, hzb_dag_fn fn, int32_t device);
/* a pre-populated state of 2^k consecutive accounts held as per-level arrays (circuits_amd/builder.py DenseState): levels[d] =
 * [2^d][32] node hashes, value = [2^k][32] state hashes, account j = (key_idx[j], mant[j] * 10^expo[j]) with token 1, nonce 0 and the
 * key (sign, ay, eth) of its owner. The arrays stay the caller's and must outlive the database. Only on an empty database. */
int hzb_db_set_base(hzb_db* db, int32_t k, uint64_t first_idx, const uint8_t* const* levels, const uint8_t* value, const uint8_t* key_idx,
                    const uint64_t* mant, const uint8_t* expo, int32_t n_keys, const uint8_t* key_sign, const uint8_t* key_ay, const uint8_t* key_eth);
/* direct state construction (what earlier deposit batches leave behind): the account gets idx = last_idx + 1 */
int hzb_db_add_account(hzb_db* db, const hzb_leaf* leaf, uint64_t* idx);
int hzb_db_get_account(hzb_db* db, uint64_t idx, hzb_leaf* out);   /* HZB_ERR_ARG when absent */
int hzb_db_state_root(hzb_db* db, uint8_t* out);                   /* evaluates pending hashes */
uint64_t hzb_db_last_idx(const hzb_db* db);
uint32_t hzb_db_num_batch(const hzb_db* db);

hzb_batch* hzb_batch_create(hzb_db* db, int32_t n_tx, int32_t n_levels, int32_t max_l1, int32_t max_fee);
void hzb_batch_destroy(hzb_batch* b);
int hzb_batch_add_tx(hzb_batch* b, const hzb_tx* tx);
int hzb_batch_add_txs(hzb_batch* b, const hzb_tx* txs, uint64_t n);   /* n calls of hzb_batch_add_tx; stops at the first refusal */
/* The synthetic benchmark batch of the reference's generator (tools/generate-input.js:61-109) into an empty batch on a database with
 * a pre-populated state: maxL1Tx deposits of random ones of the n_keys L1 keys, signed transfers of 20 % of the sender's balance
 * (the first `exits` of them exits), userFee 176, fee token 1, one fee receiver -- the transactions circuits_amd/native_builder.py
 * synthetic_batch_native makes from the same seed (CPython's random.Random(seed) restated bit for bit). signer_keys[k] = private
 * scalar of the state's owner key k (hzb_db_set_base key_idx). */
int hzb_batch_add_synthetic(hzb_batch* b, uint64_t seed, int32_t exits, int32_t n_keys, const uint8_t* l1_bjj_compressed, const uint8_t* l1_eth_addr,
                            int32_t n_signers, const uint8_t* signer_keys);
int hzb_batch_add_token(hzb_batch* b, uint32_t token_id);
int hzb_batch_add_fee_idx(hzb_batch* b, uint64_t idx);
/* Walks the batch, hashes it, and writes the circuit inputs in the packed bulk-upload format of hz_inputs_upload: signal i of the
 * circuit's input list (hz_input_name / hz_input_packed_offset / hz_input_packed_width) goes to packed + offsets[i] as [outer][inner]
 * little-endian elements of widths[i] bytes. Every signal the builder produces and the table names is written; a table name the
 * builder does not know is an error. hash_global_inputs receives the value the circuit's public output must take. */
int hzb_batch_build(hzb_batch* b, int32_t n_signals, const char* const* names, const uint64_t* offsets, const uint32_t* widths, uint8_t* packed, uint64_t packed_bytes,
                    uint8_t* hash_global_inputs);
/* The same in two halves, for a caller that builds batch after batch: _begin walks the batch (state changes, every packed input that
 * is no hash, the hash jobs) and hands the jobs to a worker thread; _finish waits for the digests and completes the packed inputs, the
 * signatures' S, the roots and hash_global_inputs. Between the two the caller may create, fill and _begin the NEXT batch -- on the
 * same database (the state the next batch sees is complete: Merkle nodes whose hash is on its way are named by job number, and the
 * next batch's jobs take those numbers as inputs) or on another one -- so the device evaluates batch N's Merkle hashes while the host
 * walks batch N + 1. `packed` and `hash_global_inputs` must stay valid until _finish returns. One batch per database may be on its
 * way while another is walked; _begin finishes an older one first, every other call on the database or on an unfinished batch finishes
 * what is outstanding (hzb_db_state_root, hzb_db_clone, hzb_batch_destroy ...). The packed bytes are those of hzb_batch_build. */
int hzb_batch_build_begin(hzb_batch* b, int32_t n_signals, const char* const* names, const uint64_t* offsets, const uint32_t* widths, uint8_t* packed,
                          uint64_t packed_bytes, uint8_t* hash_global_inputs);
int hzb_batch_build_finish(hzb_batch* b);
/* after build: roots, counters, and the exit leaves for Withdraw (reference test/withdraw.test.js:39-157) */
int hzb_batch_roots(const hzb_batch* b, uint8_t* new_state_root, uint8_t* new_exit_root, uint64_t* new_last_idx);
int hzb_batch_exit_proof(hzb_batch* b, uint64_t idx, hzb_leaf* leaf, uint8_t* siblings /* [n_levels + 1][32] */, int32_t* n_siblings);
int hzb_batch_tx_flags(const hzb_batch* b, int32_t i, int32_t* is_amount_nullified);
/* jobs hashed, DAG segments, device milliseconds (0 on the host path), seconds spent in the walk and in the evaluator */
int hzb_batch_stats(const hzb_batch* b, uint64_t* jobs, uint64_t* segments, double* device_ms, double* walk_s, double* eval_s);
/* seconds of the build spent SIGNING the transactions that carry a signer key (messages, nonces, R8): a wallet's work in production,
 * the synthetic generator's here; part of walk_s + eval_s */
double hzb_batch_sign_s(const hzb_batch* b);
/* self test: hash flushes (the unit of a build's deferred Merkle hashing) alive in this process; 0 once every database and batch is destroyed */
long hzb_live_flushes(void);

#ifdef __cplusplus
}
#endif
#endif
