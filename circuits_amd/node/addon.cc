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
#include <string>
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
#undef SYM
    return true;
}

static napi_value throw_hz(napi_env env, const char* what) {
    std::string m = std::string(what) + ": " + (api.last_error ? api.last_error() : "");
    napi_throw_error(env, nullptr, m.c_str());
    return nullptr;
}
static hz_ctx* get_ctx(napi_env env, napi_value v) {
    void* p = nullptr;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) { napi_throw_error(env, nullptr, "bad circuit handle"); return nullptr; }
    return (hz_ctx*)p;
}
static void finalize_ctx(napi_env, void* data, void*) { if (data && api.ctx_destroy) api.ctx_destroy((hz_ctx*)data); }

// create(templateId, nTx, nLevels, maxL1Tx, maxFeeTx, nInstances) -> handle
static napi_value Create(napi_env env, napi_callback_info info) {
    std::string err;
    if (!load_api(err)) { napi_throw_error(env, nullptr, err.c_str()); return nullptr; }
    size_t argc = 6;
    napi_value argv[6];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    int32_t a[6] = {0, 0, 0, 0, 0, 1};
    for (size_t i = 0; i < argc && i < 6; i++) napi_get_value_int32(env, argv[i], &a[i]);
    hz_params p;
    memset(&p, 0, sizeof p);
    p.template_id = a[0]; p.nTx = a[1]; p.nLevels = a[2]; p.maxL1Tx = a[3]; p.maxFeeTx = a[4]; p.n_instances = a[5]; p.device = 0;
    hz_ctx* c = nullptr;
    if (api.ctx_create(&p, &c) != HZ_OK) return throw_hz(env, "hz_ctx_create");
    napi_value ext;
    NAPI_OK(napi_create_external(env, c, finalize_ctx, nullptr, &ext));
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
    hz_status st;
    hz_error err;
    std::string msg;
};
static void run_execute(napi_env, void* data) {
    RunWork* w = (RunWork*)data;
    memset(&w->err, 0, sizeof w->err);
    w->st = api.witness_run(w->ctx, &w->err);
    if (w->st != HZ_OK) w->msg = api.last_error();
}
static void run_complete(napi_env env, napi_status, void* data) {
    RunWork* w = (RunWork*)data;
    napi_value result;
    if (w->st == HZ_OK) {
        napi_get_null(env, &result);
        napi_resolve_deferred(env, w->deferred, result);
    } else if (w->st == HZ_ERR_CONSTRAINT) {
        napi_create_object(env, &result);
        napi_value v;
        napi_create_int32(env, w->err.instance, &v); napi_set_named_property(env, result, "instance", v);
        napi_create_int32(env, w->err.unit, &v); napi_set_named_property(env, result, "unit", v);
        napi_create_int32(env, w->err.constraint_id, &v); napi_set_named_property(env, result, "constraintId", v);
        napi_create_string_utf8(env, api.constraint_name(w->err.constraint_id), NAPI_AUTO_LENGTH, &v); napi_set_named_property(env, result, "constraintName", v);
        void* p;
        napi_create_buffer_copy(env, 32, w->err.lhs, &p, &v); napi_set_named_property(env, result, "lhs", v);
        napi_create_buffer_copy(env, 32, w->err.rhs, &p, &v); napi_set_named_property(env, result, "rhs", v);
        napi_resolve_deferred(env, w->deferred, result);
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

static napi_value Init(napi_env env, napi_value exports) {
    const struct { const char* name; napi_callback fn; } fns[] = {
        {"create", Create}, {"setInput", SetInput}, {"clearInputs", ClearInputs}, {"run", Run}, {"witnessLen", WitnessLen},
        {"constraintEstimate", ConstraintEstimate}, {"read", Read}, {"lookup", Lookup}, {"inputNames", InputNames},
        {"symbolCount", SymbolCount}, {"symbolGet", SymbolGet}, {"deviceCount", DeviceCount}, {"version", Version}};
    for (const auto& f : fns) {
        napi_value fn;
        napi_create_function(env, f.name, NAPI_AUTO_LENGTH, f.fn, nullptr, &fn);
        napi_set_named_property(env, exports, f.name, fn);
    }
    return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
