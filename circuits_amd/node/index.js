"use strict";
/**
 * Node.js facade with the surface the reference's suites use on `require("circom").tester`
 * (reference test/rollup-main.test.js:4,52; test/helpers/helpers.js:139-155):
 *
 *   const circuit = await tester(pathOrSpec, opts);
 *   const w = await circuit.calculateWitness(input, sanityCheck);
 *   await circuit.assertOut(w, expected);
 *   await circuit.loadConstraints();  circuit.constraints.length
 *   await circuit.getSignal(w, "main.x")
 *
 * There is no circom compiler here: `pathOrSpec` is either a throw-away .circom file whose last
 * statement is `component main = Template(params);` (the shape every reference suite writes, e.g.
 * test/rollup-tx.test.js:36-39) or that statement itself; the template is dispatched to the
 * hand-written HIP kernels behind libhermez_witness.so. Requires a gfx950 GPU (no CPU fallback).
 */
const fs = require("fs");
const path = require("path");
const addon = require(path.join(__dirname, "hermez_addon.node"));

const R = BigInt("21888242871839275222246405745257275088548364400416034343698204186575808495617");

// template name -> [templateId, parameter names in the order of the circom template]
const TEMPLATES = {
    RollupMain: [0, ["nTx", "nLevels", "maxL1Tx", "maxFeeTx"]],
    RollupTx: [1, ["nLevels", "maxFeeTx"]],
    DecodeTx: [2, ["nLevels"]],
    FeeTx: [3, ["nLevels"]],
    HashState: [4, []],
    Withdraw: [5, ["nLevels"]],
    HashInputs: [6, ["nLevels", "nTx", "maxL1Tx", "maxFeeTx"]],
    DecodeFloat: [7, []],
    ComputeFee: [8, []],
    FeeAccumulator: [9, ["maxFeeTx"]],
    BalanceUpdater: [10, []],
    RollupTxStates: [11, []],
    RqTxVerifier: [12, []],
    Mux256: [13, []],
    BitsCompressed2AySign: [14, []],
    AySign2Ax: [15, []],
};

function parseMain(spec) {
    let text = spec;
    if (!/component\s+main/.test(spec) && fs.existsSync(spec)) text = fs.readFileSync(spec, "utf8");
    const m = /component\s+main\s*=\s*([A-Za-z0-9_]+)\s*\(([^)]*)\)/.exec(text);
    if (!m) throw new Error(`no "component main = Template(params)" in ${spec}`);
    const t = TEMPLATES[m[1]];
    if (!t) throw new Error(`template ${m[1]} is not part of the rollup-main witness path`);
    const args = m[2].split(",").map((x) => x.trim()).filter((x) => x.length).map(Number);
    if (args.length !== t[1].length || args.some((x) => !Number.isInteger(x))) throw new Error(`bad parameters for ${m[1]}: (${m[2]})`);
    const p = { nTx: 0, nLevels: 0, maxL1Tx: 0, maxFeeTx: 0 };
    t[1].forEach((n, i) => { p[n] = args[i]; });
    return { name: m[1], id: t[0], params: p };
}

function flatten(v, out) {
    if (Array.isArray(v)) { for (const x of v) flatten(x, out); } else out.push(v);
    return out;
}
function toFr(v) {
    let x = typeof v === "bigint" ? v : BigInt(v);
    x %= R;
    if (x < 0n) x += R;
    return x;
}
function packFr(values) {
    const buf = Buffer.alloc(32 * values.length);
    values.forEach((v, i) => {
        let x = toFr(v);
        for (let k = 0; k < 4; k++) { buf.writeBigUInt64LE(x & 0xFFFFFFFFFFFFFFFFn, 32 * i + 8 * k); x >>= 64n; }
    });
    return buf;
}
function unpackFr(buf, i) {
    let x = 0n;
    for (let k = 3; k >= 0; k--) x = (x << 64n) | buf.readBigUInt64LE(32 * i + 8 * k);
    return x;
}

class Circuit {
    constructor(main, opts) {
        this.main = main;
        this.opts = opts || {};
        const p = main.params;
        this.handle = addon.create(main.id, p.nTx, p.nLevels, p.maxL1Tx, p.maxFeeTx, 1);
        this.nVars = addon.witnessLen(this.handle);
        this.inputs = {};
        for (const d of addon.inputNames(this.handle)) this.inputs[d.name] = d.length;
        this._busy = Promise.resolve();
    }

    /** input: { signalName: Number | BigInt | decimal string | nested arrays } -> Array<BigInt>, w[0] === 1n */
    calculateWitness(input, sanityCheck) {
        // one in-flight call per Circuit (the reference's tests await every call serially)
        const run = this._busy.then(() => this._calculate(input, false));
        this._busy = run.catch(() => {});
        return run;
    }
    /** same, but returns the witness as a Buffer of 32-byte little-endian elements (snarkjs .wtns body) */
    calculateWitnessBin(input) {
        const run = this._busy.then(() => this._calculate(input, true));
        this._busy = run.catch(() => {});
        return run;
    }
    async _calculate(input, bin) {
        addon.clearInputs(this.handle);
        for (const key of Object.keys(input)) {
            if (!(key in this.inputs)) throw new Error(`Signal not found: ${key}`);
            const flat = flatten(input[key], []);
            if (flat.length !== this.inputs[key]) throw new Error(`Signal ${key}: expected ${this.inputs[key]} values, got ${flat.length}`);
            addon.setInput(this.handle, 0, key, packFr(flat));
        }
        const fail = await addon.run(this.handle);
        if (fail) {
            const e = new Error(`Constraint doesn't match ${unpackFr(fail.lhs, 0)} != ${unpackFr(fail.rhs, 0)} (${fail.constraintName}, unit ${fail.unit})`);
            e.constraint = fail;
            throw e;
        }
        const buf = addon.read(this.handle, 0, 0, this.nVars);
        if (bin) return buf;
        const w = new Array(this.nVars);
        for (let i = 0; i < this.nVars; i++) w[i] = unpackFr(buf, i);
        return w;
    }

    _index(name) {
        const idx = addon.lookup(this.handle, name);
        if (idx < 0) throw new Error(`Signal not found: ${name}`);
        return idx;
    }
    async getSignal(w, name) { return w[this._index(name)]; }

    /** expected: { name: value | array | nested } relative to main (reference test/helpers/helpers.js:143,154) */
    async assertOut(w, expected) {
        const check = (prefix, v) => {
            if (Array.isArray(v)) { v.forEach((x, i) => check(`${prefix}[${i}]`, x)); return; }
            if (v !== null && typeof v === "object" && typeof v !== "bigint") { for (const k of Object.keys(v)) check(`${prefix}.${k}`, v[k]); return; }
            const got = w[this._index(prefix)];
            if (got.toString() !== toFr(v).toString()) throw new Error(`${prefix}: expected ${toFr(v)} got ${got}`);
        };
        for (const k of Object.keys(expected)) check(`main.${k}`, expected[k]);
    }
    async loadConstraints() {
        // closed-form model of the reference (tools/circuit-constraints.js:31-75)
        this.constraints = { length: addon.constraintEstimate(this.handle) };
    }
    async loadSymbols() {
        this.symbols = {};
        const n = addon.symbolCount(this.handle);
        for (let i = 0; i < n; i++) { const s = addon.symbolGet(this.handle, i); this.symbols[s.name] = { varIdx: s.index }; }
    }
    async checkConstraints(_w) { /* every `===` is evaluated by calculateWitness itself */ }
    async release() { this.handle = null; }
}

async function tester(circomPathOrSpec, opts) {
    return new Circuit(parseMain(circomPathOrSpec), opts);
}

module.exports = { tester, Circuit, parseMain, deviceCount: addon.deviceCount, version: addon.version, R };
