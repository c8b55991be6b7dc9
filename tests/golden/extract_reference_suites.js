"use strict";
// Records what the reference's remaining eleven suites DO -- test/rollup-main-L1, decode-tx, withdraw, fee-tx, hash-inputs,
// balance-updater, compute-fee, rq-tx-verifier, lib/mux256, lib/utils-bjj, lib/hash-state (.test.js) -- by running the suite files under
// recording stand-ins for mocha, chai, circom.tester, circomlib, ffjavascript and @hermeznetwork/commonjs (none of which is on disk).
// Nothing is computed. Literal values are written down as they are; every value one of the absent packages would derive is written
// down as WHAT it is -- {"__ref": kind, "v": arguments}: an account's key material, a float40 encoding, a fee, a compressed
// transaction, a builder output such as bb.getInput().tokenID3[0] -- and Scalar arithmetic on such values as an expression tree
// {"__op": name, "a": operands}. tests/test_reference_suites.py resolves them with this repository's own builder and replays the
// scripts on the oracle and on the HIP path. Which `calculateWitness` a suite expects to throw, and the message text it matches, is
// found by running each case twice (the second time the stand-in throws where the first run saw `expect(true).to.be.equal(false)`).
//     node tests/golden/extract_reference_suites.js > tests/golden/reference_suites.json       (build container only)
const Module = require("module");
const path = require("path");
const REF = "/root/reference/test";
const realLog = console.log;
console.log = (...a) => console.error(...a);

// ---- symbolic values -------------------------------------------------------------------------------------------------------------------
const isSym = (x) => x !== null && ((typeof x === "object" && (x.__ref !== undefined || x.__op !== undefined)) || (typeof x === "function" && x.__path !== undefined));
const mark = (kind, v) => ({ __ref: kind, v: v === undefined ? null : v, toString(r) { return r === 16 ? mark("hex", this) : this; } });
const opNode = (name, ...a) => ({ __op: name, a });
// a list of bits of unknown length: the suites pad such lists with `while (bits.length < n) bits.push(0)` / unshift(0) -- the replay pads
// to the width of the signal the list is assigned to
const bitsOf = (x, order) => ({ __ref: "bits", v: { of: x, order }, length: Infinity, push() {}, unshift() {}, reverse() { return bitsOf(x, order === "lsb" ? "msb" : "lsb"); } });
function clone(o) {
    if (typeof o === "bigint") return o.toString();
    if (typeof o === "number" || typeof o === "string" || typeof o === "boolean" || o === null || o === undefined) return o === undefined ? null : o;
    if (typeof o === "function") return o.__path !== undefined ? { __ref: "bbPath", v: { bb: o.__bb, path: o.__path } } : undefined;
    if (Array.isArray(o)) return o.map(clone);
    if (Buffer.isBuffer(o)) return { __ref: "bytes", v: o.toString("hex") };
    const r = {};
    for (const k of Object.keys(o)) {
        if (k === "length" && o[k] === Infinity) continue;
        const v = clone(o[k]);
        if (v !== undefined) r[k] = v;
    }
    if (o.__signer !== undefined) r.signer = o.__signer;
    return r;
}
const big = (v) => (typeof v === "bigint" ? v : typeof v === "number" ? BigInt(v) : typeof v === "string" ? BigInt(v) : null);
const arith = (name, f) => (a, b) => { const x = isSym(a) ? null : big(a), y = isSym(b) ? null : big(b); return x !== null && y !== null ? f(x, y) : opNode(name, a, b); };
const Scalar = {
    e: (v) => (isSym(v) ? v : big(v)),
    add: arith("add", (a, b) => a + b), sub: arith("sub", (a, b) => a - b), mul: arith("mul", (a, b) => a * b),
    shl: arith("shl", (a, b) => a << b), shr: arith("shr", (a, b) => a >> b),
    isZero: (a) => (isSym(a) ? opNode("isZero", a) : big(a) === 0n),
    toString: (v, r) => (isSym(v) ? v : big(v).toString(r || 10)),
    fromString: (s, r) => (isSym(s) ? mark("fromString", [s, r || 10]) : BigInt((r === 16 && !String(s).startsWith("0x") ? "0x" : "") + s)),
    bits: (x) => {
        if (isSym(x)) return bitsOf(x, "lsb");
        const out = [];
        for (let v = big(x); v > 0n; v >>= 1n) out.push(Number(v & 1n));
        return out;
    },
    bitLength: (x) => (isSym(x) ? opNode("bitLength", x) : big(x).toString(2).length),
};

// ---- the batch builder of @hermeznetwork/commonjs, as a recorder ------------------------------------------------------------------------
let ops = [], nextBb = 0, nextDb = 0, nextW = 0, pass = 1, failAt = new Set(), messages = {}, nCalc = 0;
const pathProxy = (bb, pth) => new Proxy(function () {}, {
    get(_, k) {
        if (k === "__bb") return bb;
        if (k === "__path") return pth;
        if (k === "toJSON") return () => ({ __ref: "bbPath", v: { bb, path: pth } });
        if (k === "toString") return () => pathProxy(bb, pth);
        if (k === "length") return Infinity;
        if (k === "push" || k === "unshift") return () => {};
        if (typeof k === "symbol") return undefined;
        return pathProxy(bb, pth.concat([/^\d+$/.test(k) ? Number(k) : k]));
    },
    apply(_, __, args) { return pathProxy(bb, pth.concat([{ call: args.map(clone) }])); },
    has(_, k) { return k === "__path" || k === "__bb"; },
});
class Account {
    constructor(n) {
        this.n = n;
        this.bjjCompressed = mark("bjjCompressed", n); this.ethAddr = mark("ethAddr", n); this.ay = mark("ay", n); this.sign = mark("sign", n); this.ax = mark("ax", n);
    }
    signTx(tx) { Object.defineProperty(tx, "__signer", { value: this.n, enumerable: false, writable: true, configurable: true }); }
}
function makeBb(db, args) {
    const id = nextBb++;
    const rec = { id, currentNumBatch: db.numBatch + 1 };
    ops.push({ op: "buildBatch", db: db.id, bb: id, args: clone(args), currentNumBatch: rec.currentNumBatch });
    const own = {
        addTx(tx) { ops.push({ op: "addTx", bb: id, tx: clone(tx) }); },
        addToken(t) { ops.push({ op: "addToken", bb: id, token: clone(t) }); },
        addFeeIdx(i) { ops.push({ op: "addFeeIdx", bb: id, idx: clone(i) }); },
        async build() { ops.push({ op: "build", bb: id }); },
        __id: id, currentNumBatch: rec.currentNumBatch,
    };
    return new Proxy(own, {
        get(t, k) {
            if (k in t) return t[k];
            if (k === "then" || typeof k === "symbol") return undefined;
            if (k === "L1TxFullB" || k === "L1L2TxDataB") return NaN;   // `while (bits.length < n * bb.L1TxFullB)` : the replay pads
            return pathProxy(id, [k]);                                   // getInput().x[i], feeTotals[0], getNewStateRoot(), chainID ...
        },
        set(t, k, v) { if (!(k in t)) ops.push({ op: "setBb", bb: id, field: k, value: clone(v) }); t[k] = v; return true; },
    });
}
async function RollupDB() {
    const db = {
        id: nextDb++, numBatch: 0,
        async buildBatch(...a) { return makeBb(this, a); },
        async consolidate(bb) { this.numBatch = bb.currentNumBatch; ops.push({ op: "consolidate", db: this.id, bb: bb.__id }); },
        async getStateByIdx(idx) { return { balance: mark("stateBalance", clone(idx)) }; },
        async getExitTreeInfo(idx, numBatch) {
            const at = { db: this.id, idx: clone(idx), numBatch: clone(numBatch) };
            const st = {};
            for (const f of ["tokenID", "balance", "idx", "sign", "nonce", "ethAddr", "ay"]) st[f] = mark("exit." + f, at);
            return { found: true, state: st, siblings: Object.assign(mark("exit.siblings", at), { length: Infinity, push() {} }) };
        },
    };
    ops.push({ op: "newState", db: db.id });
    return db;
}

// ---- chai / circom -------------------------------------------------------------------------------------------------------------------
const lastCalc = () => { for (let i = ops.length - 1; i >= 0; i--) if (ops[i].op === "calculateWitness" || ops[i].op === "assertBatch") return ops[i]; return null; };
const expectStub = (v) => {
    const chain = {
        equal(x) { if (v === true && x === false) { const a = lastCalc(); if (a) a.expectFailure = true; } return chain; },
        greaterThan(x) { ops.push({ op: "expectGreaterThan", value: clone(v), than: clone(x) }); return chain; },
    };
    chain.to = chain; chain.be = chain;
    return chain;
};
let mains = [];   // `component main = ...` of every circuit file the suite writes, in order
function makeCircuit(main) {
    return {
        constraints: { length: 0 },
        async loadConstraints() {},
        async calculateWitness(input) {
            const n = nCalc++;
            const w = { id: nextW++ };
            const rec = clone(input);
            fixSigns(rec);
            ops.push({ op: "calculateWitness", main, w: w.id, input: rec });
            if (pass === 2 && failAt.has(n)) {
                const op = ops[ops.length - 1];
                op.expectFailure = true;
                throw { message: { includes(s) { op.message = s; return true; } } };
            }
            return w;
        },
        async assertOut(w, out) { const rec = clone(out); fixSigns(rec); ops.push({ op: "assertOut", w: w.id, expected: rec }); },
    };
}
const fsReal = require("fs");
const fsStub = Object.assign({}, fsReal, {
    writeFileSync(p, code) { const m = /component\s+main\s*=\s*([^;]+);/.exec(String(code)); mains.push(m ? m[1].replace(/\s+/g, "") : null); },
    unlinkSync() {},
});
let rngState = 12345;
const rnd = () => { rngState = (rngState * 1103515245 + 12345) % 2147483648; return rngState / 2147483648; };
const random = (n) => { let r = 0n; const N = BigInt(Math.floor(n)); if (N <= 1n) return 0; for (let i = 0; i < 5; i++) r = (r << 53n) + BigInt(Math.floor(rnd() * 2 ** 53)); r %= N; return N < (1n << 53n) ? Number(r) : r; };
let nPoints = 0, signOf = null;
const pointOf = (what) => [mark("point.x", what), mark("point.y", what)];
const helpers = {
    async depositTx(bb, account, tokenID, loadAmount) {
        bb.addTx({ fromIdx: 0, loadAmountF: mark("fix2Float", loadAmount), tokenID, fromBjjCompressed: account.bjjCompressed, fromEthAddr: account.ethAddr, toIdx: 0, onChain: true });
    },
    async assertBatch(bb) { ops.push({ op: "assertBatch", bb: bb.__id }); },
    async assertTxs(bb) { ops.push({ op: "assertTxs", bb: bb.__id }); },
    async assertAccountsBalances(accounts, balances) { ops.push({ op: "assertBalances", idx: accounts.map((a) => clone(a.idx)), balances: clone(balances) }); },
    random, printSignals: async () => {}, printBatchOutputs: async () => {},
};
const stubs = {
    circom: { tester: async () => makeCircuit(mains[mains.length - 1]) },
    chai: { expect: expectStub },
    fs: fsStub,
    crypto: Object.assign({}, require("crypto"), { randomBytes: (n) => mark("randomKey", nPoints++) }),
    circomlib: {
        SMTMemDB: class {}, poseidon: () => mark("poseidon"),
        babyJub: { Base8: pointOf("base8"), packPoint: (p) => ({ __ref: "packPoint", v: clone(p), get 31() { signOf = p; return 0; } }) },
        eddsa: { prv2pub: (k) => pointOf(clone(k)) },
    },
    ffjavascript: { Scalar, utils: { stringifyBigInts: (x) => x, leBuff2int: (b) => mark("leBuff2int", clone(b)) } },
    "@hermeznetwork/commonjs": {
        RollupDB, HermezAccount: Account,
        Constants: { exitIdx: 1, nullIdx: 0, firstIdx: 255, nullEthAddr: "0xffffffffffffffffffffffffffffffffffffffff" },
        float40: { fix2Float: (v) => mark("fix2Float", clone(v)), float2Fix: (v) => mark("float2Fix", clone(v)), round: (v) => mark("float40.round", clone(v)) },
        txUtils: {
            buildTxCompressedData: (tx) => mark("txCompressedData", clone(tx)), buildTxCompressedDataV2: (tx) => mark("txCompressedDataV2", clone(tx)),
            buildHashSig: (tx) => mark("hashSig", clone(tx)), encodeL2Tx: (tx, n) => mark("encodeL2Tx", [clone(tx), n]), encodeL1Tx: (tx, n) => mark("encodeL1Tx", [clone(tx), n]),
            encodeL1TxFull: (tx, n) => mark("encodeL1TxFull", [clone(tx), n]),
        },
        feeTable: { computeFee: (a, s) => mark("computeFee", [clone(a), clone(s)]), tableAdjustedFee: Array.from({ length: 256 }, (_, i) => mark("tableAdjustedFee", i)) },
        stateUtils: { hashState: (s) => mark("hashState", clone(s)) }, withdrawUtils: { hashInputsWithdraw: (i) => mark("hashInputsWithdraw", clone(i)) },
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

// the sign of a packed point is read from the top bit of its last byte (`compressedBuff[31] & 0x80`, test/lib/utils-bjj.test.js:18-24): the
// stand-in cannot answer that, it notes WHICH point was asked and the `sign` fields written next stand for "the sign of that point"
function fixSigns(o) {
    if (signOf === null || o === null || typeof o !== "object") return;
    if ("sign" in o && o.sign === 0) o.sign = { __ref: "point.sign", v: clone(signOf[0]).v };
}

const SUITES = ["rollup-main-L1.test.js", "decode-tx.test.js", "withdraw.test.js", "fee-tx.test.js", "hash-inputs.test.js", "balance-updater.test.js", "compute-fee.test.js",
    "rq-tx-verifier.test.js", "lib/mux256.test.js", "lib/utils-bjj.test.js", "lib/hash-state.test.js"];
async function runSuite(f, collect) {
    tests.length = 0; befores.length = 0; mains = [];
    delete require.cache[require.resolve(path.join(REF, f))];
    require(path.join(REF, f));
    for (const b of befores) await b.call({ timeout() {} });
    const suiteMains = mains.slice();
    let carry = [];   // withdraw.test.js keeps `inputs` from one case to the next: the cases of a suite run in one recording context
    for (const [name, fn] of tests) {
        ops = carry.slice(); nCalc = 0;
        const base = ops.length;
        let error = null;
        try { await fn.call({ timeout() {} }); } catch (e) { error = String((e && e.message) || e); }
        collect(name, ops.slice(base), error);
        carry = [];
    }
    return suiteMains;
}
async function main() {
    const out = [];
    for (const f of SUITES) {
        // pass 1: which calculateWitness calls are expected to throw; pass 2: throw there, and keep the message text the suite matches
        const first = [];
        pass = 1; failAt = new Set(); nextBb = 0; nextDb = 0; nextW = 0; rngState = 12345; nPoints = 0; signOf = null;
        await runSuite(f, (name, o) => first.push(o));
        const cases = [];
        pass = 2; nextBb = 0; nextDb = 0; nextW = 0; rngState = 12345; nPoints = 0; signOf = null;
        let ci = 0;
        // (failAt is per case: set before each case runs)
        tests.length = 0; befores.length = 0; mains = [];
        delete require.cache[require.resolve(path.join(REF, f))];
        require(path.join(REF, f));
        for (const b of befores) await b.call({ timeout() {} });
        const suiteMains = mains.slice();
        for (const [name, fn] of tests) {
            ops = []; nCalc = 0;
            failAt = new Set();
            let k = 0;
            for (const o of first[ci]) if (o.op === "calculateWitness") { if (o.expectFailure) failAt.add(k); k++; }
            let error = null;
            try { await fn.call({ timeout() {} }); } catch (e) { error = String((e && e.message) || e); }
            // batch-level failures (assertBatch + expect(true).to.be.equal(false)) are known from pass 1
            first[ci].forEach((o, i) => { if (o.op === "assertBatch" && o.expectFailure && ops[i] && ops[i].op === "assertBatch") ops[i].expectFailure = true; });
            cases.push({ suite: f, case: name, mains: suiteMains, ops, recordingError: error });
            ci++;
        }
        out.push(...cases);
    }
    realLog(JSON.stringify({ source: "scripts of eleven suites recorded from /root/reference/test (see extract_reference_suites.js); __ref / __op objects stand for values the absent JS packages derive", cases: out }));
}
main().catch((e) => { console.error(e); process.exit(1); });
