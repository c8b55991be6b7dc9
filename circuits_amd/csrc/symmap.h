// The resolved symbol map of a circom compile (formats.hip builds it, export.hip takes it to the device): per variable of the
// compiler's numbering either a stored signal of this layout, or a DERIVED variable evaluated from stored ones.
#pragma once
#include <stdint.h>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>
#include "derived.h"

enum { DV_POSEIDON = 1, DV_LINEAR = 2, DV_ISZERO_IN = 3, DV_PRODUCT = 4, DV_QUOTIENT = 5 };   // DV_PRODUCT: lins[lin] * lins[lin + 1] + lins[lin + 2]; DV_QUOTIENT: lins[lin] / lins[lin + 1] + lins[lin + 2] (0 / 0 = 0)
using hzderived::PW_ARK_IN; using hzderived::PW_ARK_OUT; using hzderived::PW_MIX_IN; using hzderived::PW_MIX_OUT;
struct DerivedVar {
    uint8_t kind = 0, t = 0, what = 0;
    uint16_t round = 0, lane = 0;
    uint32_t lin = 0;                 // DV_LINEAR: index into hz_symmap::lins
    uint64_t first = 0, stride = 0;   // DV_POSEIDON: this library's index of sigmaF[0][0].in2 and the distance between consecutive stored signals
};
static const uint64_t DERIVED_FLAG = 1ull << 63;
struct LinForm {
    hzh::F c0;
    std::vector<std::pair<hzh::F, uint64_t>> terms;   // coefficient (Montgomery form), stored index or DERIVED_FLAG | derived index
};
namespace hzexp { struct DevPlan; void devplan_free(DevPlan*); }
struct hz_symmap {
    std::vector<uint64_t> index;          // per variable: index in this library's per-instance witness, DERIVED_FLAG | k, or ~0 = unresolved
    std::vector<std::string> first_label; // per unresolved variable (in variable order): one of its names
    std::vector<uint64_t> unresolved;     // variable numbers
    std::vector<DerivedVar> derived;
    std::vector<LinForm> lins;
    std::unordered_map<std::string, uint64_t> memo; // name -> resolved index of a DERIVED signal (rules refer to each other); stored names are looked up each time
    struct PosBlk { int t; uint64_t first, stride; };
    std::map<std::string, PosBlk> pos_memo;   // component prefix -> its Poseidon block (t = 0: not one)
    uint64_t n_derived = 0, n_solved = 0;
    // the constraint system of the same compile (hz_symmap_create_r1cs): linear combination q of constraint c, q = 0..2 for A, B, C,
    // holds the terms [off[3c + q], off[3c + q + 1]) of (wire, index into the coefficient pool)
    struct R1cs {
        uint64_t n_wires = 0, n_cons = 0;
        std::vector<uint64_t> off;
        std::vector<uint32_t> wire, coef;
        std::vector<hzh::F> pool;
    } r1cs;
    // device-resident form of the map (export.hip): tables only, one plan per (device, geometry) a context has asked for -- built by the
    // first hz_witness_export_dev / hz_symmap_upload of such a context and kept until the map is destroyed (never freed or rebuilt under
    // another context's feet); what an export writes besides its output lives in the context (ctx_internal.h ExportScratch)
    mutable std::vector<hzexp::DevPlan*> devs;
    mutable std::mutex dev_mu;
    ~hz_symmap() { for (hzexp::DevPlan* d : devs) hzexp::devplan_free(d); }
};

