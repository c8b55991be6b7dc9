// N-API addon: the thin Node.js binding of include/hermez_witness.h. It mirrors what the
// reference's tests reach through `require("circom").tester` (reference test/rollup-main.test.js:4,52;
// test/helpers/helpers.js:142-154): create a circuit, set inputs by signal name, calculate the
// witness without blocking the event loop (napi_async_work), read signals by name.
// The addon dlopen()s libhermez_witness.so (C ABI only; no HIP or C++ types cross this file).
#include <dlfcn.h>
#include <node_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <string>
#include <utility>
#include <vector>
#include "../../include/hermez_witness.h"

#define NAPI_OK(call)                                             \
    do {                                                          \
        if ((call) != napi_ok) {                                  \
            napi_throw_error(env, nullptr, "N-API call failed: " #call); \
            return nullptr;                                       \
        }                                                         \
    } while (0)

struct Api {
    void* h = nullptr;
    decltype(&hz_version) version;
    decltype(&hz_last_error) last_error;
    decltype(&hz_device_count) device_count;
    decltype(&hz_ctx_create) ctx_create;
    decltype(&hz_ctx_destroy) ctx_destroy;
    decltype(&hz_witness_len) witness_len;
    decltype(&hz_constraint_estimate) constraint_estimate;
    decltype(&hz_set_input) set_input;
    decltype(&hz_clear_inputs) clear_inputs;
    decltype(&hz_input_count) input_count;
    decltype(&hz_input_name) input_name;
    decltype(&hz_witness_run) witness_run;
    decltype(&hz_witness_read) witness_read;
    decltype(&hz_symbol_count) symbol_count;
    decltype(&hz_symbol_get) symbol_get;
    decltype(&hz_symbol_lookup) symbol_lookup;
    decltype(&hz_constraint_name) constraint_name;
    // the batched path (what bench.py measures): packed inputs from pinned memory, staged uploads, enqueue / check
    decltype(&hz_inputs_packed_bytes) inputs_packed_bytes;
    decltype(&hz_input_packed_width) input_packed_width;
    decltype(&hz_input_packed_offset) input_packed_offset;
    decltype(&hz_host_alloc) host_alloc;
    decltype(&hz_host_free) host_free;
    decltype(&hz_inputs_upload) inputs_upload;
    decltype(&hz_inputs_stage) inputs_stage;
    decltype(&hz_inputs_stage_range) inputs_stage_range;
    decltype(&hz_witness_enqueue) witness_enqueue;
    decltype(&hz_witness_check) witness_check;
    decltype(&hz_witness_failures) witness_failures;
    decltype(&hz_witness_total) witness_total;
    decltype(&hz_witness_read_raw) witness_read_raw;
    decltype(&hz_witness_dev_ptr) witness_dev_ptr;
    decltype(&hz_set_inputs_json) set_inputs_json;
    decltype(&hz_witness_write_json) witness_write_json;
    decltype(&hz_witness_write_wtns) witness_write_wtns;
    decltype(&hz_symbols_write_sym) symbols_write_sym;
    decltype(&hz_symmap_create) symmap_create;
    decltype(&hz_symmap_create_r1cs) symmap_create_r1cs;
    decltype(&hz_symmap_check_r1cs) symmap_check_r1cs;
    decltype(&hz_symmap_destroy) symmap_destroy;
    decltype(&hz_symmap_nvars) symmap_nvars;
    decltype(&hz_symmap_unresolved) symmap_unresolved;
    decltype(&hz_witness_write_wtns_sym) witness_write_wtns_sym;
    decltype(&hz_witness_export_host) witness_export_host;
    decltype(&hz_symmap_upload) symmap_upload;
    decltype(&hz_symmap_solved) symmap_solved;
    decltype(&hz_symmap_derived) symmap_derived;
    decltype(&hz_poseidon_batch) poseidon_batch;
    // one batch over the GPUs of a node (hz_shard_step: the collectives live in the library)
    decltype(&hz_comm_create) comm_create;
    decltype(&hz_comm_destroy) comm_destroy;
    decltype(&hz_shard_step) shard_step;
    decltype(&hz_ctx_set_shard) ctx_set_shard;
    decltype(&hz_shard_range) shard_range;
} api;

static bool load_api(std::string& err) {
    if (api.h) return true;
    Dl_info info;
    std::string dir = ".";
    if (dladdr((void*)&load_api, &info) && info.dli_fname) {
        dir = info.dli_fname;
        dir = dir.substr(0, dir.find_last_of('/'));
    }
    const std::string path = dir + "/../libhermez_witness.so";
    api.h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!api.h) { err = std::string("cannot load ") + path + ": " + dlerror(); return false; }
#define SYM(f) api.f = (decltype(api.f))dlsym(api.h, "hz_" #f); if (!api.f) { err = "missing symbol hz_" #f; return false; }
    SYM(version) SYM(last_error) SYM(device_count) SYM(ctx_create) SYM(ctx_destroy) SYM(witness_len) SYM(constraint_estimate) SYM(set_input)
    SYM(clear_inputs) SYM(input_count) SYM(input_name) SYM(witness_run) SYM(witness_read) SYM(symbol_count) SYM(symbol_get) SYM(symbol_lookup)
    SYM(constraint_name)
    SYM(inputs_packed_bytes) SYM(input_packed_width) SYM(input_packed_offset) SYM(host_alloc) SYM(host_free) SYM(inputs_upload) SYM(inputs_stage)
    SYM(inputs_stage_range) SYM(witness_enqueue) SYM(witness_check) SYM(witness_failures) SYM(witness_total) SYM(witness_read_raw) SYM(witness_dev_ptr)
    SYM(set_inputs_json) SYM(witness_write_json) SYM(witness_write_wtns) SYM(symbols_write_sym) SYM(symmap_create) SYM(symmap_create_r1cs) SYM(symmap_check_r1cs) SYM(symmap_destroy)
    SYM(symmap_nvars) SYM(symmap_unresolved) SYM(witness_write_wtns_sym) SYM(witness_export_host) SYM(symmap_upload) SYM(symmap_solved) SYM(symmap_derived) SYM(poseidon_batch)
    SYM(comm_create) SYM(comm_destroy) SYM(shard_step) SYM(ctx_set_shard) SYM(shard_range)
#undef SYM
    return true;
}

static napi_value throw_hz(napi_env env, const char* what) {
    std::string m = std::string(what) + ": " + (api.last_error ? api.last_error() : "");
    napi_throw_error(env, nullptr, m.c_str());
    return nullptr;
}
// What a circuit handle points to: the context, and the host buffers whose asynchronous upload may still be in flight. A staged copy
// (stageRange / step / upload) is issued on a stream and consumed by the NEXT enqueue; its source ArrayBuffer must outlive the DMA,
// whatever the JS side does with it. Enqueues are numbered; a buffer staged after enqueue e is referenced until the check of
// enqueue e + 1 has completed (that enqueue waited for the copy on the device).
// "A context is used by one thread at a time" (hermez_witness.h): every work item that drives a context from the libuv pool (run, check,
// failures, step, export) holds the context's mutex, so two pending promises of one circuit never interleave inside the library
// (Promise.all([m.witnessBin(0), m.witnessBin(1)]) once shared a staging buffer between two pool threads).
static std::mutex g_ctx_mu_reg;
static std::vector<std::pair<hz_ctx*, std::mutex*>> g_ctx_mu;
static std::mutex& ctx_mu(hz_ctx* c) {
    std::lock_guard<std::mutex> g(g_ctx_mu_reg);
    for (auto& e : g_ctx_mu) if (e.first == c) return *e.second;
    g_ctx_mu.push_back({c, new std::mutex()});
    return *g_ctx_mu.back().second;
}
static void ctx_mu_forget(hz_ctx* c) {
    std::lock_guard<std::mutex> g(g_ctx_mu_reg);
    for (size_t i = 0; i < g_ctx_mu.size(); i++)
        if (g_ctx_mu[i].first == c) { delete g_ctx_mu[i].second; g_ctx_mu.erase(g_ctx_mu.begin() + (long)i); return; }
}
struct NodeCtx {
    hz_ctx* c = nullptr;
    uint64_t enq = 0;                                    // enqueues issued so far
    std::vector<std::pair<napi_ref, uint64_t>> held;     // (reference, enqueue count when it was staged); JS thread only
};
static NodeCtx* get_node(napi_env env, napi_value v) {
    void* p = nullptr;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) { napi_throw_error(env, nullptr, "bad circuit handle"); return nullptr; }
    return (NodeCtx*)p;
}
static hz_ctx* get_ctx(napi_env env, napi_value v) {
    NodeCtx* n = get_node(env, v);
    return n ? n->c : nullptr;
}
// JS thread: drop the references of every copy that an enqueue numbered <= `checked` has consumed
static void release_held(napi_env env, NodeCtx* n, uint64_t checked) {
    size_t k = 0;
    for (size_t i = 0; i < n->held.size(); i++) {
        if (n->held[i].second < checked) napi_delete_reference(env, n->held[i].first);
        else n->held[k++] = n->held[i];
    }
    n->held.resize(k);
}
static void hold(napi_env env, NodeCtx* n, napi_value buf) {
    napi_ref r = nullptr;
    if (napi_create_reference(env, buf, 1, &r) == napi_ok && r) n->held.push_back({r, n->enq});
}
static void finalize_ctx(napi_env env, void* data, void*) {
    NodeCtx* n = (NodeCtx*)data;
    if (!n) return;
    if (n->c) ctx_mu_forget(n->c);
    if (n->c && api.ctx_destroy) api.ctx_destroy(n->c);   // synchronises the device: nothing reads the held buffers after this
    for (auto& h : n->held) napi_delete_reference(env, h.first);
    delete n;
}

// create(templateId, nTx, nLevels, maxL1Tx, maxFeeTx, nInstances = 1, flags = 0, device = 0) -> handle
static napi_value Create(napi_env env, napi_callback_info info) {
    std::string err;
    if (!load_api(err)) { napi_throw_error(env, nullptr, err.c_str()); return nullptr; }
    size_t argc = 8;
    napi_value argv[8];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    int32_t a[8] = {0, 0, 0, 0, 0, 1, 0, 0};
    for (size_t i = 0; i < argc && i < 8; i++) napi_get_value_int32(env, argv[i], &a[i]);
    hz_params p;
    memset(&p, 0, sizeof p);
    p.template_id = a[0]; p.nTx = a[1]; p.nLevels = a[2]; p.maxL1Tx = a[3]; p.maxFeeTx = a[4]; p.n_instances = a[5]; p.flags = a[6]; p.device = a[7];
    hz_ctx* c = nullptr;
    if (api.ctx_create(&p, &c) != HZ_OK) return throw_hz(env, "hz_ctx_create");
    NodeCtx* n = new NodeCtx();
    n->c = c;
    napi_value ext;
    if (napi_create_external(env, n, finalize_ctx, nullptr, &ext) != napi_ok) { api.ctx_destroy(c); delete n; napi_throw_error(env, nullptr, "napi_create_external"); return nullptr; }
    return ext;
}

// setInput(handle, instance, name, Buffer of 32-byte LE values)
static napi_value SetInput(napi_env env, napi_callback_info info) {
    size_t argc = 4;
    napi_value argv[4];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    hz_ctx* c = get_ctx(env, argv[0]);
    if (!c) return nullptr;
    int32_t inst = 0;
    napi_get_value_int32(env, argv[1], &inst);
    char name[256];
    size_t nl = 0;
    NAPI_OK(napi_get_value_string_utf8(env, argv[2], name, sizeof name, &nl));
    void* data = nullptr;
    size_t len = 0;
    NAPI_OK(napi_get_buffer_info(env, argv[3], &data, &len));
    if (api.set_input(c, inst, name, (const uint8_t*)data, len / 32) != HZ_OK) return throw_hz(env, "hz_set_input");
    return nullptr;
}
static napi_value ClearInputs(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    hz_ctx* c = get_ctx(env, argv[0]);
    if (c) api.clear_inputs(c);
    return nullptr;
}

// run(handle) -> Promise<null | {instance, unit, constraintId, constraintName, lhs: Buffer, rhs: Buffer}>
struct RunWork {
    napi_async_work work;
    napi_deferred deferred;
    hz_ctx* ctx;
    NodeCtx* node = nullptr;      // check(): whose staged buffers the completed step releases
    uint64_t checked = 0;
    hz_status st;
    hz_error err;
    std::string msg;
};
static napi_value failure_record(napi_env env, const hz_error& err) {
    napi_value result, v;
    napi_create_object(env, &result);
    napi_create_int32(env, err.instance, &v); napi_set_named_property(env, result, "instance", v);
    napi_create_int32(env, err.unit, &v); napi_set_named_property(env, result, "unit", v);
    napi_create_int32(env, err.constraint_id, &v); napi_set_named_property(env, result, "constraintId", v);
    napi_create_string_utf8(env, api.constraint_name(err.constraint_id), NAPI_AUTO_LENGTH, &v); napi_set_named_property(env, result, "constraintName", v);
    void* p;
    napi_create_buffer_copy(env, 32, err.lhs, &p, &v); napi_set_named_property(env, result, "lhs", v);
    napi_create_buffer_copy(env, 32, err.rhs, &p, &v); napi_set_named_property(env, result, "rhs", v);
    return result;
}
static void run_execute(napi_env, void* data) {
    RunWork* w = (RunWork*)data;
    std::lock_guard<std::mutex> one_at_a_time(ctx_mu(w->ctx));
    memset(&w->err, 0, sizeof w->err);
    w->st = api.witness_run(w->ctx, &w->err);
    if (w->st != HZ_OK) w->msg = api.last_error();
}
static void run_complete(napi_env env, napi_status, void* data) {
    RunWork* w = (RunWork*)data;
    napi_value result;
    if (w->node && (w->st == HZ_OK || w->st == HZ_ERR_CONSTRAINT)) release_held(env, w->node, w->checked);
    if (w->st == HZ_OK) {
        napi_get_null(env, &result);
        napi_resolve_deferred(env, w->deferred, result);
    } else if (w->st == HZ_ERR_CONSTRAINT) {
        napi_resolve_deferred(env, w->deferred, failure_record(env, w->err));
    } else {
        napi_value msg, e;
        napi_create_string_utf8(env, w->msg.c_str(), NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, nullptr, msg, &e);
        napi_reject_deferred(env, w->deferred, e);
    }
    napi_delete_async_work(env, w->work);
    delete w;
}
static napi_value Run(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    hz_ctx* c = get_ctx(env, argv[0]);
    if (!c) return nullptr;
    RunWork* w = new RunWork();
    w->ctx = c;
    napi_value promise, name;
    NAPI_OK(napi_create_promise(env, &w->deferred, &promise));
    NAPI_OK(napi_create_string_utf8(env, "hz_witness_run", NAPI_AUTO_LENGTH, &name));
    NAPI_OK(napi_create_async_work(env, nullptr, name, run_execute, run_complete, w, &w->work));
    NAPI_OK(napi_queue_async_work(env, w->work));
    return promise;
}

static napi_value WitnessLen(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    hz_ctx* c = get_ctx(env, argv[0]);
    if (!c) return nullptr;
    napi_value v;
    NAPI_OK(napi_create_double(env, (double)api.witness_len(c), &v));
    return v;
}
static napi_value ConstraintEstimate(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    hz_ctx* c = get_ctx(env, argv[0]);
    if (!c) return nullptr;
    napi_value v;
    NAPI_OK(napi_create_double(env, (double)api.constraint_estimate(c), &v));
    return v;
}
// read(handle, instance, first, count) -> Buffer (count * 32 bytes)
static napi_value Read(napi_env env, napi_callback_info info) {
    size_t argc = 4;
    napi_value argv[4];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    hz_ctx* c = get_ctx(env, argv[0]);
    if (!c) return nullptr;
    int32_t inst = 0;
    double first = 0, count = 0;
    napi_get_value_int32(env, argv[1], &inst);
    napi_get_value_double(env, argv[2], &first);
    napi_get_value_double(env, argv[3], &count);
    void* data = nullptr;
    napi_value buf;
    NAPI_OK(napi_create_buffer(env, (size_t)count * 32, &data, &buf));
    if (api.witness_read(c, inst, (uint64_t)first, (uint64_t)count, (uint8_t*)data) != HZ_OK) return throw_hz(env, "hz_witness_read");
    return buf;
}
// lookup(handle, name) -> index or -1
static napi_value Lookup(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    hz_ctx* c = get_ctx(env, argv[0]);
    if (!c) return nullptr;
    char name[512];
    size_t nl = 0;
    NAPI_OK(napi_get_value_string_utf8(env, argv[1], name, sizeof name, &nl));
    uint64_t idx = 0;
    napi_value v;
    NAPI_OK(napi_create_double(env, api.symbol_lookup(c, name, &idx) ? (double)idx : -1.0, &v));
    return v;
}
static napi_value InputNames(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    hz_ctx* c = get_ctx(env, argv[0]);
    if (!c) return nullptr;
    const int32_t n = api.input_count(c);
    napi_value arr;
    NAPI_OK(napi_create_array_with_length(env, n, &arr));
    for (int32_t i = 0; i < n; i++) {
        uint64_t len = 0;
        const char* nm = api.input_name(c, i, &len);
        napi_value o, v;
        napi_create_object(env, &o);
        napi_create_string_utf8(env, nm, NAPI_AUTO_LENGTH, &v); napi_set_named_property(env, o, "name", v);
        napi_create_double(env, (double)len, &v); napi_set_named_property(env, o, "length", v);
        napi_set_element(env, arr, i, o);
    }
    return arr;
}
static napi_value SymbolCount(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    hz_ctx* c = get_ctx(env, argv[0]);
    if (!c) return nullptr;
    napi_value v;
    NAPI_OK(napi_create_double(env, (double)api.symbol_count(c), &v));
    return v;
}
static napi_value SymbolGet(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    hz_ctx* c = get_ctx(env, argv[0]);
    if (!c) return nullptr;
    double i = 0;
    napi_get_value_double(env, argv[1], &i);
    hz_symbol s;
    if (api.symbol_get(c, (uint64_t)i, &s) != HZ_OK) return throw_hz(env, "hz_symbol_get");
    napi_value o, v;
    napi_create_object(env, &o);
    napi_create_string_utf8(env, s.name, NAPI_AUTO_LENGTH, &v); napi_set_named_property(env, o, "name", v);
    napi_create_double(env, (double)s.index, &v); napi_set_named_property(env, o, "index", v);
    return o;
}
static napi_value DeviceCount(napi_env env, napi_callback_info) {
    std::string err;
    if (!load_api(err)) { napi_throw_error(env, nullptr, err.c_str()); return nullptr; }
    napi_value v;
    NAPI_OK(napi_create_int32(env, api.device_count(), &v));
    return v;
}
static napi_value Version(napi_env env, napi_callback_info) {
    std::string err;
    if (!load_api(err)) { napi_throw_error(env, nullptr, err.c_str()); return nullptr; }
    napi_value v;
    NAPI_OK(napi_create_string_utf8(env, api.version(), NAPI_AUTO_LENGTH, &v));
    return v;
}


// ---- the batched path ------------------------------------------------------------------------------------------------------------
// What the measured loop of bench.py does, for a Node host (reference tools/helpers/actions.js:132-146 runs one witness binary per
// batch; a serving process keeps contexts of many instances resident): packed inputs in pinned memory -> stageRange (async H2D)
// -> enqueue (kernels, asynchronous) -> check (Promise, waits on the libuv pool) -> the witness stays in HBM (devPtr) or goes to a
// .wtns file (writeWtns).
static bool get_args(napi_env env, napi_callback_info info, size_t want, napi_value* argv, size_t* got = nullptr) {
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok) { napi_throw_error(env, nullptr, "bad arguments"); return false; }
    for (size_t i = argc; i < want; i++) napi_get_undefined(env, &argv[i]);
    if (got) *got = argc;
    return true;
}
static double num(napi_env env, napi_value v, double dflt = 0) {
    double d = dflt;
    napi_valuetype t;
    if (napi_typeof(env, v, &t) == napi_ok && t == napi_number) napi_get_value_double(env, v, &d);
    return d;
}
// a byte offset / stride / count from JavaScript: a finite, non-negative integer (a cast of NaN or of a negative double is undefined)
static bool get_size(napi_env env, napi_value v, const char* what, size_t* out, bool has_default = false, size_t dflt = 0) {
    napi_valuetype t = napi_undefined;
    napi_typeof(env, v, &t);
    if (t == napi_undefined && has_default) { *out = dflt; return true; }
    double d = 0;
    if (t != napi_number || napi_get_value_double(env, v, &d) != napi_ok || !(d >= 0) || d > 9007199254740992.0 || d != (double)(uint64_t)d) {
        std::string m = std::string(what) + ": expected a non-negative integer";
        napi_throw_range_error(env, nullptr, m.c_str());
        return false;
    }
    *out = (size_t)d;
    return true;
}
// off + (count - 1) * stride + each <= len, without wrapping
static bool range_fits(size_t off, size_t count, size_t stride, size_t each, size_t len) {
    if (count == 0) return off <= len;
    size_t span = 0, end = 0;
    if (__builtin_mul_overflow(count - 1, stride, &span) || __builtin_add_overflow(span, each, &span) || __builtin_add_overflow(off, span, &end)) return false;
    return end <= len;
}
// bytes of an ArrayBuffer / Buffer / TypedArray argument
static bool get_bytes(napi_env env, napi_value v, uint8_t** data, size_t* len) {
    bool is = false;
    void* p = nullptr;
    if (napi_is_arraybuffer(env, v, &is) == napi_ok && is) { if (napi_get_arraybuffer_info(env, v, &p, len) != napi_ok) return false; *data = (uint8_t*)p; return true; }
    if (napi_is_buffer(env, v, &is) == napi_ok && is) { if (napi_get_buffer_info(env, v, &p, len) != napi_ok) return false; *data = (uint8_t*)p; return true; }
    if (napi_is_typedarray(env, v, &is) == napi_ok && is) {
        napi_typedarray_type tt; size_t n; napi_value ab; size_t off;
        if (napi_get_typedarray_info(env, v, &tt, &n, &p, &ab, &off) != napi_ok) return false;
        static const size_t w[] = {1, 1, 1, 2, 2, 4, 4, 4, 8, 8, 8};
        *data = (uint8_t*)p; *len = n * w[tt];
        return true;
    }
    napi_throw_error(env, nullptr, "expected an ArrayBuffer, Buffer or TypedArray");
    return false;
}
// packedLayout(handle) -> { bytes, inputs: [{ name, length, offset, width }] }
static napi_value PackedLayout(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    hz_ctx* c = get_ctx(env, argv[0]);
    if (!c) return nullptr;
    napi_value o, arr, v;
    napi_create_object(env, &o);
    napi_create_double(env, (double)api.inputs_packed_bytes(c), &v); napi_set_named_property(env, o, "bytes", v);
    const int32_t n = api.input_count(c);
    napi_create_array_with_length(env, n, &arr);
    for (int32_t i = 0; i < n; i++) {
        uint64_t len = 0;
        const char* nm = api.input_name(c, i, &len);
        napi_value e;
        napi_create_object(env, &e);
        napi_create_string_utf8(env, nm, NAPI_AUTO_LENGTH, &v); napi_set_named_property(env, e, "name", v);
        napi_create_double(env, (double)len, &v); napi_set_named_property(env, e, "length", v);
        napi_create_double(env, (double)api.input_packed_offset(c, i), &v); napi_set_named_property(env, e, "offset", v);
        napi_create_int32(env, api.input_packed_width(c, i), &v); napi_set_named_property(env, e, "width", v);
        napi_set_element(env, arr, i, e);
    }
    napi_set_named_property(env, o, "inputs", arr);
    return o;
}
static void finalize_pinned(napi_env, void* data, void*) { if (data && api.host_free) api.host_free(data); }
// hostAlloc(bytes) -> ArrayBuffer over pinned host memory (freed with the ArrayBuffer)
static napi_value HostAlloc(napi_env env, napi_callback_info info) {
    std::string err;
    if (!load_api(err)) { napi_throw_error(env, nullptr, err.c_str()); return nullptr; }
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    const size_t bytes = (size_t)num(env, argv[0]);
    void* p = bytes ? api.host_alloc(bytes) : nullptr;
    if (!p) return throw_hz(env, "hz_host_alloc");
    napi_value ab;
    if (napi_create_external_arraybuffer(env, p, bytes, finalize_pinned, nullptr, &ab) != napi_ok) { api.host_free(p); napi_throw_error(env, nullptr, "external ArrayBuffer"); return nullptr; }
    return ab;
}
// upload(handle, instance, bytes, byteOffset = 0): packed inputs of one instance, copy + scatter on the context's stream
static napi_value Upload(napi_env env, napi_callback_info info) {
    napi_value argv[4];
    if (!get_args(env, info, 4, argv)) return nullptr;
    NodeCtx* n = get_node(env, argv[0]);
    uint8_t* data; size_t len, off, inst;
    if (!n || !get_bytes(env, argv[2], &data, &len) || !get_size(env, argv[1], "upload: instance", &inst) || !get_size(env, argv[3], "upload: byteOffset", &off, true, 0)) return nullptr;
    const size_t each = (size_t)api.inputs_packed_bytes(n->c);
    if (inst > 0x7fffffff || !range_fits(off, 1, each, each, len)) { napi_throw_range_error(env, nullptr, "upload: buffer too small"); return nullptr; }
    if (api.inputs_upload(n->c, (int32_t)inst, data + off, each, nullptr) != HZ_OK) return throw_hz(env, "hz_inputs_upload");
    hold(env, n, argv[2]);
    return nullptr;
}
// stageRange(handle, first, count, bytes, byteOffset = 0, stride = packed bytes): the next step's inputs, asynchronous H2D
static napi_value StageRange(napi_env env, napi_callback_info info) {
    napi_value argv[6];
    if (!get_args(env, info, 6, argv)) return nullptr;
    NodeCtx* n = get_node(env, argv[0]);
    uint8_t* data; size_t len, first, count, off, stride;
    if (!n || !get_bytes(env, argv[3], &data, &len)) return nullptr;
    const size_t each = (size_t)api.inputs_packed_bytes(n->c);
    if (!get_size(env, argv[1], "stageRange: first", &first) || !get_size(env, argv[2], "stageRange: count", &count) ||
        !get_size(env, argv[4], "stageRange: byteOffset", &off, true, 0) || !get_size(env, argv[5], "stageRange: stride", &stride, true, each)) return nullptr;
    if (first > 0x7fffffff || count > 0x7fffffff || !range_fits(off, count, stride, each, len)) { napi_throw_range_error(env, nullptr, "stageRange: buffer too small"); return nullptr; }
    if (api.inputs_stage_range(n->c, (int32_t)first, (int32_t)count, data + off, each, stride, nullptr) != HZ_OK) return throw_hz(env, "hz_inputs_stage_range");
    if (count) hold(env, n, argv[3]);
    return nullptr;
}
// enqueue(handle): the kernels of one step on the context's stream; returns at once
static napi_value Enqueue(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    NodeCtx* n = get_node(env, argv[0]);
    if (!n) return nullptr;
    if (api.witness_enqueue(n->c, nullptr) != HZ_OK) return throw_hz(env, "hz_witness_enqueue");
    n->enq++;
    return nullptr;
}
// check(handle) -> Promise<null | failure record> (same shape as run): waits for the step on the libuv pool
static void check_execute(napi_env, void* data) {
    RunWork* w = (RunWork*)data;
    std::lock_guard<std::mutex> one_at_a_time(ctx_mu(w->ctx));
    memset(&w->err, 0, sizeof w->err);
    w->st = api.witness_check(w->ctx, &w->err);
    if (w->st != HZ_OK) w->msg = api.last_error();
}
static napi_value Check(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    NodeCtx* n = get_node(env, argv[0]);
    if (!n) return nullptr;
    RunWork* w = new RunWork();
    w->ctx = n->c; w->node = n; w->checked = n->enq;
    napi_value promise, name;
    NAPI_OK(napi_create_promise(env, &w->deferred, &promise));
    NAPI_OK(napi_create_string_utf8(env, "hz_witness_check", NAPI_AUTO_LENGTH, &name));
    NAPI_OK(napi_create_async_work(env, nullptr, name, check_execute, run_complete, w, &w->work));
    NAPI_OK(napi_queue_async_work(env, w->work));
    return promise;
}
// checkSync(handle) -> null | failure record: hz_witness_check on the calling (JS) thread. Blocks the event loop for as long as the
// step takes: for command-line tools and measurements, not for servers.
static napi_value CheckSync(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    NodeCtx* n = get_node(env, argv[0]);
    if (!n) return nullptr;
    hz_error err;
    memset(&err, 0, sizeof err);
    const hz_status st = api.witness_check(n->c, &err);
    napi_value result;
    if (st == HZ_OK || st == HZ_ERR_CONSTRAINT) release_held(env, n, n->enq);
    if (st == HZ_OK) { napi_get_null(env, &result); return result; }
    if (st != HZ_ERR_CONSTRAINT) return throw_hz(env, "hz_witness_check");
    return failure_record(env, err);
}
// failures(handle) -> Promise<[failure record]>: the first violated constraint of EVERY instance of the step that check() reported
// on (hz_witness_failures: a launch evaluates nInstances circuits, the reference one per call), ordered by instance
struct FailWork {
    napi_async_work work;
    napi_deferred deferred;
    hz_ctx* ctx;
    hz_status st;
    std::vector<hz_error> recs;
    std::string msg;
};
static void fail_execute(napi_env, void* data) {
    FailWork* w = (FailWork*)data;
    std::lock_guard<std::mutex> one_at_a_time(ctx_mu(w->ctx));
    size_t n = 0;
    w->st = api.witness_failures(w->ctx, nullptr, 0, &n);
    if (w->st == HZ_OK && n) {
        w->recs.resize(n);
        w->st = api.witness_failures(w->ctx, w->recs.data(), n, &n);
        if (w->st == HZ_OK) w->recs.resize(n);
    }
    if (w->st != HZ_OK) w->msg = api.last_error();
}
static void fail_complete(napi_env env, napi_status, void* data) {
    FailWork* w = (FailWork*)data;
    if (w->st == HZ_OK) {
        napi_value arr;
        napi_create_array_with_length(env, w->recs.size(), &arr);
        for (size_t i = 0; i < w->recs.size(); i++) napi_set_element(env, arr, (uint32_t)i, failure_record(env, w->recs[i]));
        napi_resolve_deferred(env, w->deferred, arr);
    } else {
        napi_value msg, e;
        napi_create_string_utf8(env, w->msg.c_str(), NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, nullptr, msg, &e);
        napi_reject_deferred(env, w->deferred, e);
    }
    napi_delete_async_work(env, w->work);
    delete w;
}
static napi_value Failures(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    hz_ctx* c = get_ctx(env, argv[0]);
    if (!c) return nullptr;
    FailWork* w = new FailWork();
    w->ctx = c;
    napi_value promise, name;
    NAPI_OK(napi_create_promise(env, &w->deferred, &promise));
    NAPI_OK(napi_create_string_utf8(env, "hz_witness_failures", NAPI_AUTO_LENGTH, &name));
    NAPI_OK(napi_create_async_work(env, nullptr, name, fail_execute, fail_complete, w, &w->work));
    NAPI_OK(napi_queue_async_work(env, w->work));
    return promise;
}
// devPtr(handle) -> BigInt device address of the physical witness buffer (for a prover in the same process); witnessTotal -> elements
static napi_value DevPtr(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    hz_ctx* c = get_ctx(env, argv[0]);
    if (!c) return nullptr;
    napi_value v;
    NAPI_OK(napi_create_bigint_uint64(env, (uint64_t)(uintptr_t)api.witness_dev_ptr(c), &v));
    return v;
}
static napi_value WitnessTotal(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    hz_ctx* c = get_ctx(env, argv[0]);
    if (!c) return nullptr;
    napi_value v;
    NAPI_OK(napi_create_double(env, (double)api.witness_total(c), &v));
    return v;
}
// readRaw(handle, first, count) -> Buffer: the physical (signal-major) buffer
static napi_value ReadRaw(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return nullptr;
    hz_ctx* c = get_ctx(env, argv[0]);
    if (!c) return nullptr;
    const uint64_t first = (uint64_t)num(env, argv[1]), count = (uint64_t)num(env, argv[2]);
    void* data = nullptr;
    napi_value buf;
    NAPI_OK(napi_create_buffer(env, (size_t)count * 32, &data, &buf));
    if (api.witness_read_raw(c, first, count, (uint8_t*)data) != HZ_OK) return throw_hz(env, "hz_witness_read_raw");
    return buf;
}
static bool get_str(napi_env env, napi_value v, std::string& out) {
    size_t n = 0;
    if (napi_get_value_string_utf8(env, v, nullptr, 0, &n) != napi_ok) return false;
    out.resize(n);
    return napi_get_value_string_utf8(env, v, &out[0], n + 1, &n) == napi_ok;
}
// setInputsJson(handle, instance, text): the input.json of the reference's tools (tools/generate-input.js:109)
static napi_value SetInputsJson(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return nullptr;
    hz_ctx* c = get_ctx(env, argv[0]);
    std::string text;
    if (!c || !get_str(env, argv[2], text)) return nullptr;
    if (api.set_inputs_json(c, (int32_t)num(env, argv[1]), text.c_str(), text.size()) != HZ_OK) return throw_hz(env, "hz_set_inputs_json");
    return nullptr;
}
// writeWtns(handle, instance, path[, symText[, r1cs Buffer[, check]]]) / writeJson(handle, instance, path) / writeSym(handle, path)
// symText: the .sym of the circom compile -> the compiler's variable order; r1cs: the .r1cs of the same compile -> the variables no
// label resolves are solved from its linear constraints (hz_symmap_create_r1cs); check: every constraint is evaluated first and a
// witness that violates one is not written (hz_symmap_check_r1cs)
static napi_value WriteWtns(napi_env env, napi_callback_info info) {
    napi_value argv[6];
    size_t got = 0;
    if (!get_args(env, info, 6, argv, &got)) return nullptr;
    hz_ctx* c = get_ctx(env, argv[0]);
    std::string path, sym;
    if (!c || !get_str(env, argv[2], path)) return nullptr;
    const int32_t inst = (int32_t)num(env, argv[1]);
    napi_valuetype t = napi_undefined;
    if (got >= 4) napi_typeof(env, argv[3], &t);
    if (got >= 4 && t == napi_string) {
        if (!get_str(env, argv[3], sym)) return nullptr;
        hz_symmap* m = nullptr;
        bool is_buf = false, check = false;
        void* rdata = nullptr; size_t rlen = 0;
        if (got >= 5 && napi_is_buffer(env, argv[4], &is_buf) == napi_ok && is_buf) napi_get_buffer_info(env, argv[4], &rdata, &rlen);
        if (got >= 6) napi_get_value_bool(env, argv[5], &check);
        if (is_buf) {
            if (api.symmap_create_r1cs(c, sym.c_str(), sym.size(), (const uint8_t*)rdata, rlen, &m) != HZ_OK) return throw_hz(env, "hz_symmap_create_r1cs");
        } else if (api.symmap_create(c, sym.c_str(), sym.size(), &m) != HZ_OK) return throw_hz(env, "hz_symmap_create");
        uint64_t var = 0; const char* nm = nullptr;
        const uint64_t miss = api.symmap_unresolved(m, 0, &var, &nm);
        if (miss) {
            std::string e = "circom .sym: " + std::to_string(miss) + " variables are not stored by this layout (first: " + (nm ? nm : "?") + ")";
            api.symmap_destroy(m);
            napi_throw_error(env, nullptr, e.c_str());
            return nullptr;
        }
        if (check && is_buf) {
            uint64_t n_bad = 0, first = 0;
            const hz_status cs = api.symmap_check_r1cs(c, m, inst, &n_bad, &first, 1);
            if (cs != HZ_OK) { api.symmap_destroy(m); return throw_hz(env, "hz_symmap_check_r1cs"); }
            if (n_bad) {
                std::string e = std::to_string(n_bad) + " constraints of the .r1cs do not hold on this witness (first: constraint " + std::to_string(first) + ")";
                api.symmap_destroy(m);
                napi_throw_error(env, nullptr, e.c_str());
                return nullptr;
            }
        }
        const hz_status st = api.witness_write_wtns_sym(c, m, inst, path.c_str());
        api.symmap_destroy(m);
        if (st != HZ_OK) return throw_hz(env, "hz_witness_write_wtns_sym");
        return nullptr;
    }
    if (api.witness_write_wtns(c, inst, path.c_str()) != HZ_OK) return throw_hz(env, "hz_witness_write_wtns");
    return nullptr;
}
// ---- a circom .sym (+ .r1cs) imported ONCE and kept: the witness in the compiler's variable order -------------------------------------------
// importSym(handle, symText, r1cs Buffer | null) -> map handle (an external; freed with the handle's finaliser or freeMap)
// mapInfo(map) -> { nVars, unresolved, firstUnresolved, solved, derived }
// exportWitness(handle, map, instance) -> Promise<ArrayBuffer of nVars x 32 bytes>: hz_witness_export_host on the libuv pool (one device pass
//                                         + D2H; what the reference's calculateWitness returns, test/helpers/helpers.js:142,149, as bytes)
// writeWtnsMap(handle, map, instance, file), checkMap(handle, map, instance) -> { bad, first }
struct NodeMap {
    hz_symmap* m = nullptr;
    int in_flight = 0;          // export work items queued or running (JS thread only): freeMap waits for them
    bool free_pending = false;
};
static void map_finalize(napi_env, void* data, void*) {   // (a pending work item holds a reference to the external: never runs under one)
    NodeMap* nm = (NodeMap*)data;
    if (nm->m) api.symmap_destroy(nm->m);
    delete nm;
}
static NodeMap* get_map(napi_env env, napi_value v) {
    void* p = nullptr;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p || !((NodeMap*)p)->m || ((NodeMap*)p)->free_pending) { napi_throw_error(env, nullptr, "bad or released symbol-map handle"); return nullptr; }
    return (NodeMap*)p;
}
static napi_value ImportSym(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    size_t got = 0;
    if (!get_args(env, info, 3, argv, &got)) return nullptr;
    hz_ctx* c = get_ctx(env, argv[0]);
    std::string sym;
    if (!c || !get_str(env, argv[1], sym)) return nullptr;
    bool is_buf = false;
    void* rdata = nullptr; size_t rlen = 0;
    if (got >= 3 && napi_is_buffer(env, argv[2], &is_buf) == napi_ok && is_buf) napi_get_buffer_info(env, argv[2], &rdata, &rlen);
    hz_symmap* m = nullptr;
    if (is_buf) {
        if (api.symmap_create_r1cs(c, sym.c_str(), sym.size(), (const uint8_t*)rdata, rlen, &m) != HZ_OK) return throw_hz(env, "hz_symmap_create_r1cs");
    } else if (api.symmap_create(c, sym.c_str(), sym.size(), &m) != HZ_OK) return throw_hz(env, "hz_symmap_create");
    NodeMap* nm = new NodeMap();
    nm->m = m;
    napi_value ext;
    if (napi_create_external(env, nm, map_finalize, nullptr, &ext) != napi_ok) { map_finalize(env, nm, nullptr); return nullptr; }
    return ext;
}
static napi_value MapInfo(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    NodeMap* nm = get_map(env, argv[0]);
    if (!nm) return nullptr;
    uint64_t var = 0; const char* name = nullptr;
    const uint64_t miss = api.symmap_unresolved(nm->m, 0, &var, &name);
    napi_value o, v;
    napi_create_object(env, &o);
    napi_create_double(env, (double)api.symmap_nvars(nm->m), &v); napi_set_named_property(env, o, "nVars", v);
    napi_create_double(env, (double)miss, &v); napi_set_named_property(env, o, "unresolved", v);
    napi_create_double(env, (double)api.symmap_solved(nm->m), &v); napi_set_named_property(env, o, "solved", v);
    napi_create_double(env, (double)api.symmap_derived(nm->m), &v); napi_set_named_property(env, o, "derived", v);
    if (miss && name) { napi_create_string_utf8(env, name, NAPI_AUTO_LENGTH, &v); napi_set_named_property(env, o, "firstUnresolved", v); }
    return o;
}
static napi_value FreeMap(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    void* p = nullptr;
    if (napi_get_value_external(env, argv[0], &p) == napi_ok && p && ((NodeMap*)p)->m) {
        NodeMap* nm = (NodeMap*)p;
        if (nm->in_flight > 0) nm->free_pending = true;   // an export is reading it on the pool: freed by the completion of the last one
        else { api.symmap_destroy(nm->m); nm->m = nullptr; }
    }
    return nullptr;
}
struct ExportWork {
    napi_async_work work;
    napi_deferred deferred;
    hz_ctx* ctx; hz_symmap* map;
    NodeCtx* node; NodeMap* nmap;
    napi_ref keep_ctx, keep_map;   // the circuit and map externals: neither is finalised while the work item is pending
    int32_t inst;
    uint64_t nvars;
    uint8_t* data;        // the backing store of the ArrayBuffer the promise resolves with: created on the JS thread, referenced (keep_ab)
                          // until the work item completes, written by the pool thread (no external buffer: node 12 asserts in
                          // ArrayBufferReference::Finalize when an environment ends with one alive)
    napi_ref keep_ab;
    hz_status st;
    std::string msg;
};
static void export_execute(napi_env, void* data) {
    ExportWork* w = (ExportWork*)data;
    std::lock_guard<std::mutex> one_at_a_time(ctx_mu(w->ctx));   // the export stages through the context's buffers on its main stream
    w->st = api.witness_export_host(w->ctx, w->map, w->inst, 0, w->nvars, w->data);
    if (w->st != HZ_OK) w->msg = api.last_error();
}
static void export_complete(napi_env env, napi_status, void* data) {
    ExportWork* w = (ExportWork*)data;
    if (w->st == HZ_OK) {
        napi_value ab;
        if (napi_get_reference_value(env, w->keep_ab, &ab) == napi_ok && ab) napi_resolve_deferred(env, w->deferred, ab);
        else { napi_value msg, e; napi_create_string_utf8(env, "cannot wrap the exported witness", NAPI_AUTO_LENGTH, &msg); napi_create_error(env, nullptr, msg, &e); napi_reject_deferred(env, w->deferred, e); }
    } else {
        napi_value msg, e;
        napi_create_string_utf8(env, w->msg.c_str(), NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, nullptr, msg, &e);
        napi_reject_deferred(env, w->deferred, e);
    }
    if (w->keep_ab) napi_delete_reference(env, w->keep_ab);
    if (--w->nmap->in_flight == 0 && w->nmap->free_pending && w->nmap->m) { api.symmap_destroy(w->nmap->m); w->nmap->m = nullptr; w->nmap->free_pending = false; }
    if (w->keep_ctx) napi_delete_reference(env, w->keep_ctx);
    if (w->keep_map) napi_delete_reference(env, w->keep_map);
    napi_delete_async_work(env, w->work);
    delete w;
}
static napi_value ExportWitness(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return nullptr;
    NodeCtx* node = get_node(env, argv[0]);
    hz_ctx* c = node ? node->c : nullptr;
    NodeMap* nm = get_map(env, argv[1]);
    if (!c || !nm) return nullptr;
    if (api.symmap_unresolved(nm->m, 0, nullptr, nullptr)) { napi_throw_error(env, nullptr, "the symbol map has unresolved variables (mapInfo)"); return nullptr; }
    ExportWork* w = new ExportWork();
    w->ctx = c; w->map = nm->m; w->node = node; w->nmap = nm; w->keep_ctx = w->keep_map = nullptr;
    napi_create_reference(env, argv[0], 1, &w->keep_ctx);
    napi_create_reference(env, argv[1], 1, &w->keep_map);
    nm->in_flight++; w->inst = (int32_t)num(env, argv[2]); w->nvars = api.symmap_nvars(nm->m); w->data = nullptr; w->keep_ab = nullptr; w->st = HZ_OK;
    {
        napi_value ab;
        void* store = nullptr;
        if (napi_create_arraybuffer(env, (size_t)std::max<uint64_t>(w->nvars, 1) * 32, &store, &ab) != napi_ok || !store || napi_create_reference(env, ab, 1, &w->keep_ab) != napi_ok) {
            nm->in_flight--;
            if (w->keep_ctx) napi_delete_reference(env, w->keep_ctx);
            if (w->keep_map) napi_delete_reference(env, w->keep_map);
            delete w;
            napi_throw_error(env, nullptr, "out of memory for the exported witness");
            return nullptr;
        }
        w->data = (uint8_t*)store;
    }
    napi_value promise, name;
    NAPI_OK(napi_create_promise(env, &w->deferred, &promise));
    NAPI_OK(napi_create_string_utf8(env, "hz_witness_export_host", NAPI_AUTO_LENGTH, &name));
    NAPI_OK(napi_create_async_work(env, nullptr, name, export_execute, export_complete, w, &w->work));
    NAPI_OK(napi_queue_async_work(env, w->work));
    return promise;
}
static napi_value WriteWtnsMap(napi_env env, napi_callback_info info) {
    napi_value argv[4];
    if (!get_args(env, info, 4, argv)) return nullptr;
    hz_ctx* c = get_ctx(env, argv[0]);
    NodeMap* nm = get_map(env, argv[1]);
    std::string path;
    if (!c || !nm || !get_str(env, argv[3], path)) return nullptr;
    std::lock_guard<std::mutex> one_at_a_time(ctx_mu(c));   // (an export of this circuit may be pending on the pool)
    if (api.witness_write_wtns_sym(c, nm->m, (int32_t)num(env, argv[2]), path.c_str()) != HZ_OK) return throw_hz(env, "hz_witness_write_wtns_sym");
    return nullptr;
}
static napi_value CheckMap(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return nullptr;
    hz_ctx* c = get_ctx(env, argv[0]);
    NodeMap* nm = get_map(env, argv[1]);
    if (!c || !nm) return nullptr;
    uint64_t n_bad = 0, first = 0;
    std::lock_guard<std::mutex> one_at_a_time(ctx_mu(c));
    if (api.symmap_check_r1cs(c, nm->m, (int32_t)num(env, argv[2]), &n_bad, &first, 1) != HZ_OK) return throw_hz(env, "hz_symmap_check_r1cs");
    napi_value o, v;
    napi_create_object(env, &o);
    napi_create_double(env, (double)n_bad, &v); napi_set_named_property(env, o, "bad", v);
    napi_create_double(env, n_bad ? (double)first : -1.0, &v); napi_set_named_property(env, o, "first", v);
    return o;
}
static napi_value WriteJson(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return nullptr;
    hz_ctx* c = get_ctx(env, argv[0]);
    std::string path;
    if (!c || !get_str(env, argv[2], path)) return nullptr;
    if (api.witness_write_json(c, (int32_t)num(env, argv[1]), path.c_str()) != HZ_OK) return throw_hz(env, "hz_witness_write_json");
    return nullptr;
}
static napi_value WriteSym(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    hz_ctx* c = get_ctx(env, argv[0]);
    std::string path;
    if (!c || !get_str(env, argv[1], path)) return nullptr;
    if (api.symbols_write_sym(c, path.c_str()) != HZ_OK) return throw_hz(env, "hz_symbols_write_sym");
    return nullptr;
}
// poseidonBatch(t, inputs: Buffer of n*(t-1) elements, withWitness = false, device = 0) -> { out: Buffer, witness: Buffer | null }
static napi_value PoseidonBatch(napi_env env, napi_callback_info info) {
    std::string err;
    if (!load_api(err)) { napi_throw_error(env, nullptr, err.c_str()); return nullptr; }
    napi_value argv[4];
    if (!get_args(env, info, 4, argv)) return nullptr;
    const int32_t t = (int32_t)num(env, argv[0]);
    uint8_t* in; size_t len;
    if (!get_bytes(env, argv[1], &in, &len)) return nullptr;
    if (t < 2 || t > 7 || len % (32 * (size_t)(t - 1))) { napi_throw_error(env, nullptr, "poseidonBatch: t in 2..7, inputs a multiple of 32*(t-1) bytes"); return nullptr; }
    bool wit = false;
    napi_get_value_bool(env, argv[2], &wit);
    const size_t n = len / (32 * (size_t)(t - 1));
    static const int rp[] = {56, 57, 56, 60, 60, 63};
    const size_t nsbox = 8 * (size_t)t + rp[t - 2];
    void *po = nullptr, *pw = nullptr;
    napi_value out, w, o;
    NAPI_OK(napi_create_buffer(env, 32 * n, &po, &out));
    if (wit) NAPI_OK(napi_create_buffer(env, 96 * nsbox * n, &pw, &w));
    else napi_get_null(env, &w);
    if (api.poseidon_batch((int32_t)num(env, argv[3]), t, n, in, (uint8_t*)po, (uint8_t*)pw) != HZ_OK) return throw_hz(env, "hz_poseidon_batch");
    napi_create_object(env, &o);
    napi_set_named_property(env, o, "out", out);
    napi_set_named_property(env, o, "witness", w);
    return o;
}

// step(handle, bytes | null, byteOffset, first, count, stride) -> Promise<null | failure record of the PREVIOUS step>
// One iteration of a serving loop as ONE work item on the libuv pool: wait for and check the step enqueued before (if any), enqueue
// the next one (which scatters the inputs staged for it), stage the inputs of the one after (asynchronous H2D beside the kernels).
// A context driven this way never bounces between the JS thread and the pool inside an iteration, and two contexts are two
// independent promise chains on two pool threads (tests/node/bench_facade.js).
struct StepWork {
    napi_async_work work;
    napi_deferred deferred;
    napi_ref keep;          // the staged ArrayBuffer: handed to the context's list when the work item completes
    hz_ctx* ctx;
    NodeCtx* node;
    uint64_t checked, staged_at;
    uint8_t* data;
    size_t each, stride;
    int32_t first, count;
    bool had_prev;
    bool issued = false;    // the next step was enqueued (not when the previous one was rejected)
    hz_status st;
    hz_error err;
    std::string msg;
};
static void step_execute(napi_env, void* data) {
    StepWork* w = (StepWork*)data;
    std::lock_guard<std::mutex> one_at_a_time(ctx_mu(w->ctx));
    memset(&w->err, 0, sizeof w->err);
    w->st = HZ_OK;
    w->checked = 0;
    if (w->had_prev) {
        w->st = api.witness_check(w->ctx, &w->err);
        if (w->st != HZ_OK && w->st != HZ_ERR_CONSTRAINT) { w->msg = api.last_error(); return; }
        w->checked = w->node->enq;   // one work item per circuit at a time (index.js): nobody else moves the counter
        // a rejected step ends the pipeline here: the next enqueue would reset the failure record, and failures() -- every
        // instance's first violated constraint -- is for exactly this moment. Nothing is enqueued, nothing staged.
        if (w->st == HZ_ERR_CONSTRAINT) return;
    }
    w->issued = true;
    hz_status e = api.witness_enqueue(w->ctx, nullptr);
    if (e == HZ_OK) w->node->enq++;
    w->staged_at = w->node->enq;
    if (e == HZ_OK && w->data && w->count > 0) e = api.inputs_stage_range(w->ctx, w->first, w->count, w->data, w->each, w->stride, nullptr);
    if (e != HZ_OK) { w->st = e; w->msg = api.last_error(); }
}
static void step_complete(napi_env env, napi_status, void* data) {
    StepWork* w = (StepWork*)data;
    napi_value result;
    release_held(env, w->node, w->checked);
    if (w->keep && w->issued) w->node->held.push_back({w->keep, w->staged_at});   // released when the step that consumes the copy has been checked
    else if (w->keep) napi_delete_reference(env, w->keep);                          // nothing was staged from it
    if (w->st == HZ_OK) {
        napi_get_null(env, &result);
        napi_resolve_deferred(env, w->deferred, result);
    } else if (w->st == HZ_ERR_CONSTRAINT) {
        napi_resolve_deferred(env, w->deferred, failure_record(env, w->err));
    } else {
        napi_value msg, e;
        napi_create_string_utf8(env, w->msg.c_str(), NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, nullptr, msg, &e);
        napi_reject_deferred(env, w->deferred, e);
    }
    napi_delete_async_work(env, w->work);
    delete w;
}
static napi_value Step(napi_env env, napi_callback_info info) {
    napi_value argv[7];
    if (!get_args(env, info, 7, argv)) return nullptr;
    NodeCtx* n = get_node(env, argv[0]);
    if (!n) return nullptr;
    StepWork* w = new StepWork();
    w->ctx = n->c; w->node = n; w->keep = nullptr; w->data = nullptr; w->count = 0; w->first = 0; w->checked = 0; w->staged_at = 0;
    w->each = (size_t)api.inputs_packed_bytes(n->c);
    w->stride = w->each;
    napi_valuetype t = napi_undefined;
    napi_typeof(env, argv[1], &t);
    if (t == napi_object) {
        uint8_t* data; size_t len, off, first, count;
        if (!get_bytes(env, argv[1], &data, &len) || !get_size(env, argv[2], "step: byteOffset", &off, true, 0) || !get_size(env, argv[3], "step: first", &first, true, 0) ||
            !get_size(env, argv[4], "step: count", &count, true, 0) || !get_size(env, argv[5], "step: stride", &w->stride, true, w->each)) { delete w; return nullptr; }
        if (first > 0x7fffffff || count > 0x7fffffff || !range_fits(off, count, w->stride, w->each, len)) { delete w; napi_throw_range_error(env, nullptr, "step: buffer too small"); return nullptr; }
        w->first = (int32_t)first; w->count = (int32_t)count;
        w->data = data + off;
        if (count) napi_create_reference(env, argv[1], 1, &w->keep);
    }
    bool prev = false;
    napi_get_value_bool(env, argv[6], &prev);
    w->had_prev = prev;
    napi_value promise, name;
    NAPI_OK(napi_create_promise(env, &w->deferred, &promise));
    NAPI_OK(napi_create_string_utf8(env, "hz_step", NAPI_AUTO_LENGTH, &name));
    NAPI_OK(napi_create_async_work(env, nullptr, name, step_execute, step_complete, w, &w->work));
    NAPI_OK(napi_queue_async_work(env, w->work));
    return promise;
}

// ---- one batch over the GPUs of a node: one Node process per GPU, each with a RollupMain circuit of ONE instance -------------------------
// commCreate(transport: "rccl" | "socket", rank, world, rendezvousPath, device = 0) -> comm handle   (hz_comm_create: blocks until every rank has joined)
// shardStep(handle, comm) -> Promise<null | failure record>: hz_shard_step on the context's own stream + hz_witness_check, on the libuv pool
// setShard(handle, first, count, tail)   hz_ctx_set_shard for a host that drives the pieces itself (count < 0: back to the whole batch)
// shardRange(nTx, world, rank) -> [first, count]
struct NodeComm { hz_comm* c = nullptr; };
static void comm_finalize(napi_env, void* data, void*) {
    NodeComm* nc = (NodeComm*)data;
    if (nc->c) api.comm_destroy(nc->c);
    delete nc;
}
static napi_value CommCreate(napi_env env, napi_callback_info info) {
    std::string err;
    if (!load_api(err)) { napi_throw_error(env, nullptr, err.c_str()); return nullptr; }
    size_t argc = 5;
    napi_value argv[5];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    std::string transport, path;
    if (argc < 4 || !get_str(env, argv[0], transport) || !get_str(env, argv[3], path)) { napi_throw_error(env, nullptr, "commCreate(transport, rank, world, path, device)"); return nullptr; }
    const int32_t kind = transport == "rccl" ? HZ_COMM_RCCL : transport == "socket" ? HZ_COMM_SOCKET : 0;
    int32_t rank = 0, world = 1, device = 0;
    napi_get_value_int32(env, argv[1], &rank); napi_get_value_int32(env, argv[2], &world);
    if (argc >= 5) napi_get_value_int32(env, argv[4], &device);
    NodeComm* nc = new NodeComm();
    if (api.comm_create(kind, device, rank, world, path.empty() ? nullptr : path.c_str(), &nc->c) != HZ_OK) { delete nc; return throw_hz(env, "hz_comm_create"); }
    napi_value ext;
    if (napi_create_external(env, nc, comm_finalize, nullptr, &ext) != napi_ok) { comm_finalize(env, nc, nullptr); return nullptr; }
    return ext;
}
struct ShardWork {
    napi_async_work work;
    napi_deferred deferred;
    napi_ref keep_ctx, keep_comm;
    hz_ctx* ctx; hz_comm* comm;
    hz_status st;
    hz_error err;
    std::string msg;
};
static void shard_execute(napi_env, void* data) {
    ShardWork* w = (ShardWork*)data;
    std::lock_guard<std::mutex> one_at_a_time(ctx_mu(w->ctx));
    memset(&w->err, 0, sizeof w->err);
    w->st = api.shard_step(w->ctx, w->comm, nullptr);
    if (w->st == HZ_OK) w->st = api.witness_check(w->ctx, &w->err);
    if (w->st != HZ_OK) w->msg = api.last_error();
}
static void shard_complete(napi_env env, napi_status, void* data) {
    ShardWork* w = (ShardWork*)data;
    napi_value result;
    if (w->st == HZ_OK) { napi_get_null(env, &result); napi_resolve_deferred(env, w->deferred, result); }
    else if (w->st == HZ_ERR_CONSTRAINT) napi_resolve_deferred(env, w->deferred, failure_record(env, w->err));
    else {
        napi_value msg, e;
        napi_create_string_utf8(env, w->msg.c_str(), NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, nullptr, msg, &e);
        napi_reject_deferred(env, w->deferred, e);
    }
    if (w->keep_ctx) napi_delete_reference(env, w->keep_ctx);
    if (w->keep_comm) napi_delete_reference(env, w->keep_comm);
    napi_delete_async_work(env, w->work);
    delete w;
}
static napi_value ShardStep(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    hz_ctx* c = get_ctx(env, argv[0]);
    void* p = nullptr;
    if (!c || napi_get_value_external(env, argv[1], &p) != napi_ok || !p || !((NodeComm*)p)->c) { napi_throw_error(env, nullptr, "shardStep(circuit handle, comm)"); return nullptr; }
    ShardWork* w = new ShardWork();
    w->ctx = c; w->comm = ((NodeComm*)p)->c; w->st = HZ_OK; w->keep_ctx = w->keep_comm = nullptr;
    napi_create_reference(env, argv[0], 1, &w->keep_ctx);
    napi_create_reference(env, argv[1], 1, &w->keep_comm);
    napi_value promise, name;
    NAPI_OK(napi_create_promise(env, &w->deferred, &promise));
    NAPI_OK(napi_create_string_utf8(env, "hz_shard_step", NAPI_AUTO_LENGTH, &name));
    NAPI_OK(napi_create_async_work(env, nullptr, name, shard_execute, shard_complete, w, &w->work));
    NAPI_OK(napi_queue_async_work(env, w->work));
    return promise;
}
static napi_value SetShard(napi_env env, napi_callback_info info) {
    napi_value argv[4];
    if (!get_args(env, info, 4, argv)) return nullptr;
    hz_ctx* c = get_ctx(env, argv[0]);
    if (!c) return nullptr;
    int32_t first = 0, count = -1, tail = 1;
    napi_get_value_int32(env, argv[1], &first); napi_get_value_int32(env, argv[2], &count); napi_get_value_int32(env, argv[3], &tail);
    if (api.ctx_set_shard(c, first, count, tail) != HZ_OK) return throw_hz(env, "hz_ctx_set_shard");
    return nullptr;
}
static napi_value ShardRange(napi_env env, napi_callback_info info) {
    std::string err;
    if (!load_api(err)) { napi_throw_error(env, nullptr, err.c_str()); return nullptr; }
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return nullptr;
    int32_t nTx = 0, world = 1, rank = 0, first = 0, count = 0;
    napi_get_value_int32(env, argv[0], &nTx); napi_get_value_int32(env, argv[1], &world); napi_get_value_int32(env, argv[2], &rank);
    api.shard_range(nTx, world, rank, &first, &count);
    napi_value arr, v;
    napi_create_array_with_length(env, 2, &arr);
    napi_create_int32(env, first, &v); napi_set_element(env, arr, 0, v);
    napi_create_int32(env, count, &v); napi_set_element(env, arr, 1, v);
    return arr;
}

static napi_value Init(napi_env env, napi_value exports) {
    const struct { const char* name; napi_callback fn; } fns[] = {
        {"create", Create}, {"setInput", SetInput}, {"clearInputs", ClearInputs}, {"run", Run}, {"witnessLen", WitnessLen},
        {"constraintEstimate", ConstraintEstimate}, {"read", Read}, {"lookup", Lookup}, {"inputNames", InputNames},
        {"symbolCount", SymbolCount}, {"symbolGet", SymbolGet}, {"deviceCount", DeviceCount}, {"version", Version},
        {"packedLayout", PackedLayout}, {"hostAlloc", HostAlloc}, {"upload", Upload}, {"stageRange", StageRange}, {"enqueue", Enqueue},
        {"check", Check}, {"devPtr", DevPtr}, {"witnessTotal", WitnessTotal}, {"readRaw", ReadRaw}, {"setInputsJson", SetInputsJson},
        {"writeWtns", WriteWtns}, {"writeJson", WriteJson}, {"writeSym", WriteSym}, {"poseidonBatch", PoseidonBatch}, {"step", Step}, {"checkSync", CheckSync}, {"failures", Failures},
        {"importSym", ImportSym}, {"mapInfo", MapInfo}, {"freeMap", FreeMap}, {"exportWitness", ExportWitness}, {"writeWtnsMap", WriteWtnsMap}, {"checkMap", CheckMap},
        {"commCreate", CommCreate}, {"shardStep", ShardStep}, {"setShard", SetShard}, {"shardRange", ShardRange}};
    for (const auto& f : fns) {
        napi_value fn;
        napi_create_function(env, f.name, NAPI_AUTO_LENGTH, f.fn, nullptr, &fn);
        napi_set_named_property(env, exports, f.name, fn);
    }
    return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
