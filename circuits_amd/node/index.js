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
// A step runs on a dozen HIP streams per circuit (signature ladders, fee chain, SHA-256 tail beside the hash / tree chains); the
// runtime maps all streams onto 4 hardware queues unless told otherwise (read once, when the runtime initialises: before the addon
// makes its first call). Measured with two circuits in flight: 1.42 M tx/s with the default, 1.61 M with 16.
if (!process.env.GPU_MAX_HW_QUEUES) process.env.GPU_MAX_HW_QUEUES = "16";
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
    SMTProcessor: [16, ["nLevels"]],
    SMTVerifier: [17, ["nLevels"]],
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

function constraintError(fail) {
    const e = new Error(`Constraint doesn't match ${unpackFr(fail.lhs, 0)} != ${unpackFr(fail.rhs, 0)} (${fail.constraintName}, instance ${fail.instance}, unit ${fail.unit})`);
    e.constraint = fail;
    return e;
}
// circom_tester's second argument: `true`, or an options object whose sanityCheck defaults to on (every reference call passes a
// truthy value, test/helpers/helpers.js:142,149); only an explicit false / { sanityCheck: false } turns the constraint asserts off
function wantsSanityCheck(arg) {
    if (arg === false) return false;
    if (arg !== null && typeof arg === "object" && arg.sanityCheck === false) return false;
    return true;
}

class Circuit {
    /** opts.nInstances: independent instances of the main component evaluated by one run (default 1 = the reference's shape);
     *  opts.flags: 2 = HZ_FLAG_LATENCY (one to four batches per circuit: its kernel chains on compute units of their own; two such circuits
     *  in flight overlap, plain ones do not), 6 = with HZ_FLAG_SOLO (nothing else runs on the device: 7.9 ms for one 2048-transaction
     *  batch; include/hermez_witness.h); opts.device: HIP device ordinal */
    constructor(main, opts) {
        this.main = main;
        this.opts = opts || {};
        const p = main.params;
        this.nInstances = Math.max(1, this.opts.nInstances | 0);
        this.handle = addon.create(main.id, p.nTx, p.nLevels, p.maxL1Tx, p.maxFeeTx, this.nInstances, this.opts.flags | 0, this.opts.device | 0);
        this.nVars = addon.witnessLen(this.handle);
        this.inputs = {};
        for (const d of addon.inputNames(this.handle)) this.inputs[d.name] = d.length;
        this._busy = Promise.resolve();
    }
    _serial(fn) {
        // one in-flight call per Circuit (the reference's tests await every call serially)
        const run = this._busy.then(fn);
        this._busy = run.catch(() => {});
        return run;
    }

    /** input: { signalName: Number | BigInt | decimal string | nested arrays } -> Array<BigInt>, w[0] === 1n.
     *  sanityCheck (circom_tester's second argument): constraint asserts on unless explicitly false -- then a violated
     *  constraint does not reject and the (complete) witness is returned as computed. */
    calculateWitness(input, sanityCheck) {
        return this._serial(() => this._calculate(input, false, wantsSanityCheck(sanityCheck)));
    }
    /** same, but returns the witness as a Buffer of 32-byte little-endian elements (snarkjs .wtns body) */
    calculateWitnessBin(input, sanityCheck) {
        return this._serial(() => this._calculate(input, true, wantsSanityCheck(sanityCheck)));
    }
    _setInputs(instance, input) {
        for (const key of Object.keys(input)) {
            if (!(key in this.inputs)) throw new Error(`Signal not found: ${key}`);
            const flat = flatten(input[key], []);
            if (flat.length !== this.inputs[key]) throw new Error(`Signal ${key}: expected ${this.inputs[key]} values, got ${flat.length}`);
            addon.setInput(this.handle, instance, key, packFr(flat));
        }
    }
    async _calculate(input, bin, sanity) {
        addon.clearInputs(this.handle);
        this._setInputs(0, input);
        const fail = await addon.run(this.handle);
        if (fail && sanity) throw constraintError(fail);
        const buf = addon.read(this.handle, 0, 0, this.nVars);
        if (bin) return buf;
        const w = new Array(this.nVars);
        for (let i = 0; i < this.nVars; i++) w[i] = unpackFr(buf, i);
        return w;
    }

    // ---- many instances per run: the path the benchmark measures (bench.py), for a Node host ---------------------------------
    /** inputs: one input object per instance (nInstances of them). Resolves to a reader { get(instance, name), bin(instance) }:
     *  a 2048-transaction RollupMain witness is 120 M elements -- it stays on the device (devPtr) unless asked for. */
    calculateWitnessBatch(inputs, sanityCheck) {
        return this._serial(async () => {
            if (!Array.isArray(inputs) || inputs.length !== this.nInstances) throw new Error(`expected ${this.nInstances} input objects`);
            addon.clearInputs(this.handle);
            inputs.forEach((inp, k) => this._setInputs(k, inp));
            const fail = await addon.run(this.handle);
            if (fail && wantsSanityCheck(sanityCheck)) throw constraintError(fail);
            return this.reader();
        });
    }
    reader() {
        return {
            get: (instance, name) => unpackFr(addon.read(this.handle, instance, this._index(name), 1), 0),
            bin: (instance) => addon.read(this.handle, instance, 0, this.nVars),
        };
    }
    /** { bytes, inputs: [{ name, length, offset, width }] }: the packed input buffer of ONE instance (include/hermez_witness.h,
     *  hz_inputs_upload): every input signal at its offset, elements of `width` bytes (32, or 1 for bit-valued signals) */
    packedLayout() {
        if (!this._layout) this._layout = addon.packedLayout(this.handle);
        return this._layout;
    }
    /** pinned host memory as an ArrayBuffer: the source of asynchronous uploads */
    hostAlloc(bytes) { return addon.hostAlloc(bytes); }
    /** marshal one input object into the packed layout at `byteOffset` of `target` (ArrayBuffer | Buffer | Uint8Array) */
    packInput(input, target, byteOffset) {
        const lay = this.packedLayout();
        const u8 = target instanceof ArrayBuffer ? new Uint8Array(target) : new Uint8Array(target.buffer, target.byteOffset, target.byteLength);
        const base = byteOffset | 0;
        if (base + lay.bytes > u8.length) throw new Error("packInput: target too small");
        const seen = new Set();
        for (const key of Object.keys(input)) {
            if (!(key in this.inputs)) throw new Error(`Signal not found: ${key}`);
            seen.add(key);
        }
        for (const d of lay.inputs) {
            if (!seen.has(d.name)) throw new Error(`Not all inputs have been set (${d.name})`);
            const flat = flatten(input[d.name], []);
            if (flat.length !== d.length) throw new Error(`Signal ${d.name}: expected ${d.length} values, got ${flat.length}`);
            if (d.width === 1) {
                // bit-valued signals travel as one byte each. A value that is not a bit must reach the device as it is -- the circuit's
                // own boolean constraint rejects it there (src/rollup-main.circom:214-216), exactly as the reference calculator would --
                // so nothing is masked; what a byte cannot carry is refused here instead of being coerced into a passing witness.
                for (let i = 0; i < flat.length; i++) {
                    const v = toFr(flat[i]);
                    if (v > 255n) throw new RangeError(`Signal ${d.name}[${i}] = ${v}: the packed form carries 0..255 per element; use calculateWitness() for this input`);
                    u8[base + d.offset + i] = Number(v);
                }
            } else {
                const b = packFr(flat);
                u8.set(b, base + d.offset);
            }
        }
    }
    // offsets / strides are validated by the addon (finite non-negative integers): `undefined` takes the default, nothing else is coerced
    upload(instance, buf, byteOffset) { addon.upload(this.handle, instance, buf, byteOffset === undefined ? 0 : byteOffset); }
    stageRange(first, count, buf, byteOffset, stride) {
        addon.stageRange(this.handle, first, count, buf, byteOffset === undefined ? 0 : byteOffset, stride === undefined || stride === 0 ? this.packedLayout().bytes : stride);
    }
    /** the kernels of one step, asynchronous; check() resolves when they are done and rejects on the first violated constraint */
    enqueue() { this._inFlight = true; addon.enqueue(this.handle); }
    async check(sanityCheck) {
        this._inFlight = false;
        const fail = await addon.check(this.handle);
        if (fail && wantsSanityCheck(sanityCheck)) throw constraintError(fail);
    }
    /** One iteration of a serving loop in one hop to the thread pool: check the previous step of this circuit (if one is in flight),
     *  enqueue the next, stage the inputs of the one after from `buf` (omit buf: nothing staged). Resolves when the enqueue has
     *  been issued; rejects if the PREVIOUS step violated a constraint -- and then enqueues and stages NOTHING, so that failures()
     *  still describes the rejected launch; call step() again to go on. Finish a loop with check(). */
    async step(buf, byteOffset, first, count, stride, sanityCheck) {
        const hadPrev = !!this._inFlight;
        const work = addon.step(this.handle, buf || null, byteOffset === undefined ? 0 : byteOffset, first === undefined ? 0 : first, buf ? (count === undefined ? 0 : count) : 0,
                                stride === undefined || stride === 0 ? this.packedLayout().bytes : stride, hadPrev);   // bad arguments throw here: nothing enqueued
        this._inFlight = true;
        const fail = await work;
        if (fail) this._inFlight = false;   // a rejected step is not followed: nothing was enqueued or staged, failures() can be asked now
        if (fail && wantsSanityCheck(sanityCheck)) throw constraintError(fail);
    }
    /** check() on the calling thread: blocks the event loop until the step is done (command-line tools, measurements) */
    checkSync(sanityCheck) {
        this._inFlight = false;
        const fail = addon.checkSync(this.handle);
        if (fail && wantsSanityCheck(sanityCheck)) throw constraintError(fail);
    }
    /** After a check() / step() that reported a violated constraint: the first violated constraint of EVERY instance of that launch
     *  (the reference evaluates one circuit per calculateWitness call; a launch here evaluates nInstances), ordered by instance,
     *  each as the `constraint` record of the error check() throws. Empty when the launch was clean. */
    async failures() {
        return (await addon.failures(this.handle)).map((f) => Object.assign({ message: constraintError(f).message }, f));
    }
    // ---- one batch over the GPUs of a node: one Node process per GPU, each with a RollupMain circuit of one instance --------------------
    // (the reference's counterpart: the `-n` thread-per-component mode of its compiled witness calculator, tools/helpers/actions.js:39-45)
    /** Joins the communicator of this rank (hz_comm_create; blocks until every rank has): transport "rccl" (RCCL over xGMI, loaded by the
     *  library with dlopen) or "socket" (the two small collectives staged through host memory over the rendezvous: no RCCL needed);
     *  `path`: a Unix socket name every rank can reach -- rank 0 listens, the others connect. */
    joinComm(transport, rank, world, path) { this._comm = addon.commCreate(transport, rank | 0, world | 0, path || "", this.opts.device | 0); return this._comm; }
    /** Every rank sets the SAME inputs (the whole batch) and calls this: the rank's transaction range, the all_gather of the
     *  data-availability records, rank 0's FeeTx / message / SHA-256 chain, the broadcast, the rank's share of the block witness, and the
     *  check of this rank's constraints. The witness stays sharded: getSignal / readRaw serve what this rank computed. */
    shardStep(input, sanityCheck) {
        return this._serial(async () => {
            if (!this._comm) throw new Error("shardStep: joinComm(transport, rank, world, path) first");
            if (input) { addon.clearInputs(this.handle); this._setInputs(0, input); }
            const fail = await addon.shardStep(this.handle, this._comm);
            if (fail && wantsSanityCheck(sanityCheck)) throw constraintError(fail);
        });
    }
    /** for a host that drives the pieces itself: this circuit evaluates transactions [first, first + count) only (count < 0: all again) */
    setShard(first, count, tail) { addon.setShard(this.handle, first | 0, count | 0, tail ? 1 : 0); }
    static shardRange(nTx, world, rank) { return addon.shardRange(nTx | 0, world | 0, rank | 0); }
    /** one signal of instance 0 by name, straight from the device (a sharded circuit holds only its own part of the witness) */
    readSignal(name) { return unpackFr(addon.read(this.handle, 0, this._index(name), 1), 0); }
    devPtr() { return addon.devPtr(this.handle); }
    witnessTotal() { return addon.witnessTotal(this.handle); }
    readRaw(first, count) { return addon.readRaw(this.handle, first, count); }
    setInputsJson(text, instance) { addon.setInputsJson(this.handle, instance | 0, text); }
    /** snarkjs .wtns of one instance; with `symText` (a circom .sym) in the compiler's variable order; with `r1cs` (a Buffer: the .r1cs
     *  of the same compile) the wire-through variables of an unreduced compile are solved from its linear constraints, and with
     *  `check` every constraint is evaluated first (an Error instead of a file when one does not hold) */
    writeWtns(file, instance, symText, r1cs, check) {
        if (symText && r1cs) addon.writeWtns(this.handle, instance | 0, file, symText, r1cs, !!check);
        else if (symText) addon.writeWtns(this.handle, instance | 0, file, symText);
        else addon.writeWtns(this.handle, instance | 0, file);
    }
    /** A circom .sym (and, for a compile without constraint reduction, its .r1cs as a Buffer) imported ONCE: the object that turns this
     *  circuit's witnesses into the compiler's variable order -- what the reference's calculateWitness returns and its prove step reads
     *  (test/helpers/helpers.js:142,149, tools/helpers/actions.js:132-170). Throws when a variable cannot be served. */
    importSym(symText, r1cs) {
        const map = addon.importSym(this.handle, symText, r1cs || null);
        const info = addon.mapInfo(map);
        if (info.unresolved) { addon.freeMap(map); throw new Error(`circom .sym: ${info.unresolved} variables are not stored by this layout (first: ${info.firstUnresolved})`); }
        const circuit = this;
        return {
            nVars: info.nVars, solved: info.solved, derived: info.derived,
            /** w[0 .. nVars) of one instance as bytes (32-byte little-endian elements): one device pass + one copy, off the event loop */
            witnessBin(instance) { return addon.exportWitness(circuit.handle, map, instance | 0); },
            /** the same as BigInt[] (tests; a full-size witness does not fit a JS array) */
            async witness(instance) {
                const u = new BigUint64Array(await addon.exportWitness(circuit.handle, map, instance | 0));
                const w = new Array(u.length / 4);
                for (let i = 0; i < w.length; i++) w[i] = u[4 * i] | (u[4 * i + 1] << 64n) | (u[4 * i + 2] << 128n) | (u[4 * i + 3] << 192n);
                return w;
            },
            writeWtns(file, instance) { addon.writeWtnsMap(circuit.handle, map, instance | 0, file); },
            /** every constraint of the .r1cs on the exported witness: { bad, first } (first = -1 when none) */
            check(instance) { return addon.checkMap(circuit.handle, map, instance | 0); },
            release() { addon.freeMap(map); },
        };
    }
    writeJson(file, instance) { addon.writeJson(this.handle, instance | 0, file); }
    writeSym(file) { addon.writeSym(this.handle, file); }

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

/** n Poseidon permutations of width t on the device: rows of t-1 inputs -> digests (BigInt[]); with `witness` also the S-box
 *  signals as a Buffer of [3*(8t+RP)][n] elements (circomlib Poseidon(nInputs), reference src/lib/hash-state.circom:32) */
function poseidonBatch(t, rows, witness, device) {
    const flat = flatten(rows, []);
    const r = addon.poseidonBatch(t, packFr(flat), !!witness, device | 0);
    const n = r.out.length / 32;
    const out = new Array(n);
    for (let i = 0; i < n; i++) out[i] = unpackFr(r.out, i);
    return witness ? { out, witness: r.witness } : out;
}

module.exports = { tester, Circuit, parseMain, poseidonBatch, hostAlloc: addon.hostAlloc, deviceCount: addon.deviceCount, version: addon.version, R };
