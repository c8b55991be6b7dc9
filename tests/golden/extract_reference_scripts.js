"use strict";
// Records the SCENARIO SCRIPTS the reference's suites hold -- which batches are built, which transactions (with the suites' own
// literal field values) go into them, who signs, what is consolidated, which batch / transaction is asserted and which call the
// suite expects to fail with "Constraint doesn't match" -- by running the suite files under recording stand-ins for mocha,
// chai, circom.tester, @hermeznetwork/commonjs (RollupDB / BatchBuilder / HermezAccount) and the suites' own helper module.
// Nothing is computed: the reference's JS packages are not on disk, so expected roots and hashes cannot be recorded; the
// scripts are replayed in Python on this repository's batch builder, oracle and HIP path (tests/test_reference_scripts.py).
// Run in the build container only:   node tests/golden/extract_reference_scripts.js > tests/golden/reference_scripts.json
// Suites: test/rollup-tx.test.js:56-919, test/rollup-main.test.js:65-900 (the other eleven suites: extract_reference_suites.js).
const Module = require("module");
const path = require("path");
const REF = "/root/reference/test";
const realLog = console.log;
console.log = (...a) => console.error(...a);

let ops = [];            // of the running test case
let nextBb = 0, nextDb = 0;
const clone = (o) => JSON.parse(JSON.stringify(o, (k, v) => {
    if (typeof v === "bigint") return v.toString();
    if (k === "__signer") return undefined;
    return v;
}));
const mark = (kind, v) => ({ __ref: kind, v });

class Account {
    constructor(n) {
        this.n = n;
        this.bjjCompressed = mark("bjjCompressed", n); this.ethAddr = mark("ethAddr", n); this.ay = mark("ay", n); this.sign = mark("sign", n);
    }
    signTx(tx) { Object.defineProperty(tx, "__signer", { value: this.n, enumerable: false, writable: true, configurable: true }); }
}
const txSnapshot = (tx) => { const t = clone(tx); if (tx.__signer !== undefined) t.signer = tx.__signer; return t; };

function makeBb(db, args) {
    const id = nextBb++;
    const bb = {
        id, currentNumBatch: db.numBatch + 1, txs: [], maxNTx: args[0], totalFeeTransactions: args[3],
        addTx(tx) { this.txs.push(tx); ops.push({ op: "addTx", bb: id, tx: txSnapshot(tx) }); },
        addToken(t) { ops.push({ op: "addToken", bb: id, token: clone(t) }); },
        addFeeIdx(i) { ops.push({ op: "addFeeIdx", bb: id, idx: clone(i) }); },
        async build() { ops.push({ op: "build", bb: id }); },
        getInput() { return { __bb: id, maxNumBatch: {}, __overrides: true }; },
        getHashInputs() { return mark("hashInputs", id); },
    };
    ops.push({ op: "buildBatch", db: db.id, bb: id, args: clone(args), currentNumBatch: bb.currentNumBatch });
    return bb;
}
async function RollupDB() {
    const db = {
        id: nextDb++, numBatch: 0,
        async buildBatch(...a) { return makeBb(this, a); },
        async consolidate(bb) { this.numBatch = bb.currentNumBatch; ops.push({ op: "consolidate", db: this.id, bb: bb.id }); },
        async getStateByIdx(idx) { return { balance: 0 }; },
        async getExitTreeInfo(idx, numBatch) { ops.push({ op: "getExitTreeInfo", idx: clone(idx), numBatch: clone(numBatch) }); return { state: mark("exitState", clone(idx)), siblings: [] }; },
    };
    ops.push({ op: "newState", db: db.id });
    return db;
}
const lastAssert = () => { for (let i = ops.length - 1; i >= 0; i--) if (ops[i].op.startsWith("assert") || ops[i].op === "calculateWitness") return ops[i]; return null; };
const expectStub = (v) => {
    const chain = { equal(x) { if (v === true && x === false) { const a = lastAssert(); if (a) a.expectFailure = true; } return chain; } };
    chain.to = chain; chain.be = chain;
    return chain;
};
const circuit = {
    constraints: { length: 0 },
    async loadConstraints() {},
    async calculateWitness(input) {
        if (input && input.__bb !== undefined) {
            const ov = {};
            for (const k of Object.keys(input)) if (!k.startsWith("__") && Object.keys(input[k]).length) ov[k] = clone(input[k]);
            ops.push({ op: "calculateWitness", bb: input.__bb, overrides: ov });
        } else if (input && input.__single !== undefined) {
            const ov = {};
            for (const k of Object.keys(input)) if (k !== "__single") ov[k] = clone(input[k]);   // e.g. res.input.tokenID1 = 2 (rollup-tx.test.js:912)
            ops.push({ op: "calculateWitness", bb: input.__single.bb, tx: input.__single.tx, overrides: ov });
        } else ops.push({ op: "calculateWitness", literal: clone(input) });
        return {};
    },
    async assertOut() {},
};
const helpers = {
    async depositTx(bb, account, tokenID, loadAmount) {
        bb.addTx({ fromIdx: 0, loadAmountF: mark("fix2Float", loadAmount), tokenID, fromBjjCompressed: account.bjjCompressed, fromEthAddr: account.ethAddr, toIdx: 0, onChain: true });
    },
    getSingleTxInput(bb, numTx) { return { input: { __single: { bb: bb.id, tx: numTx } }, output: {} }; },
    async assertTxs(bb) { ops.push({ op: "assertTxs", bb: bb.id }); },
    async assertBatch(bb) { ops.push({ op: "assertBatch", bb: bb.id }); },
    async assertAccountsBalances(accounts, balances) { ops.push({ op: "assertBalances", idx: accounts.map((a) => clone(a.idx)), balances: clone(balances) }); },
    random: () => 0, printSignals: async () => {}, printBatchOutputs: async () => {},
};
const Scalar = { e: (v) => v, add: (a, b) => a + b, toString: (v) => String(v), shl: (a, b) => mark("shl", [clone(a), clone(b)]), fromString: (s, r) => mark("fromString", [s, r || 10]) };
const fsReal = require("fs");
const stubs = {
    circom: { tester: async () => circuit },
    chai: { expect: expectStub },
    fs: Object.assign({}, fsReal, { writeFileSync() {}, unlinkSync() {} }),
    circomlib: { SMTMemDB: class {}, poseidon: () => 0 },
    ffjavascript: { Scalar, utils: { stringifyBigInts: (x) => x } },
    "@hermeznetwork/commonjs": {
        RollupDB, HermezAccount: Account,
        Constants: { exitIdx: 1, nullIdx: 0, firstIdx: 255, nullEthAddr: "0xffffffffffffffffffffffffffffffffffffffff" },
        float40: { fix2Float: (v) => mark("fix2Float", clone(v)), float2Fix: (v) => v },
        txUtils: { buildTxCompressedDataV2: (tx) => mark("txCompressedDataV2", txSnapshot(tx)), buildHashSig: () => 0 },
        feeTable: {}, stateUtils: { hashState: (s) => mark("hashState", clone(s)) }, withdrawUtils: { hashInputsWithdraw: (i) => mark("hashInputsWithdraw", clone(i)) },
        utils: { padZeros: (s) => s, extract: () => 0 },
    },
};
const origLoad = Module._load;
Module._load = function (request) {
    if (Object.prototype.hasOwnProperty.call(stubs, request)) return stubs[request];
    if (request.endsWith("helpers/helpers")) return helpers;
    return origLoad.apply(this, arguments);
};
const tests = [], befores = [];
global.describe = (name, fn) => { fn.call({ timeout() {} }); };
global.it = (name, fn) => tests.push([name, fn]);
global.before = (fn) => befores.push(fn);
global.after = () => {};

async function main() {
    const out = [];
    for (const f of ["rollup-tx.test.js", "rollup-main.test.js"]) {
        tests.length = 0; befores.length = 0;
        require(path.join(REF, f));
        for (const b of befores) await b.call({ timeout() {} });
        for (const [name, fn] of tests) {
            ops = []; nextBb = 0; nextDb = 0;
            let error = null;
            try { await fn.call({ timeout() {} }); } catch (e) { error = String(e && e.message || e); }
            out.push({ suite: f, case: name, ops, recordingError: error });
        }
    }
    realLog(JSON.stringify({ source: "scenario scripts recorded from /root/reference/test (see extract_reference_scripts.js); __ref objects stand for values the absent JS packages derive (account keys, float40 encodings, hashes)", cases: out }));
}
main().catch((e) => { console.error(e); process.exit(1); });
