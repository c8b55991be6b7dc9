"use strict";
// The measured loop of bench.py from a Node.js host (VERDICT r2 item 4): RollupMain contexts of B instances, `inflight` of them,
// every step = stageRange (packed inputs of the step's B batches from pinned memory, asynchronous H2D) + enqueue + check, through
// the N-API addon. Inputs: a file of `n` packed batches written by bench.py (circuits_amd/batchgen.py built them).
// usage: node bench_facade.js <packed.bin> <nTx> <nLevels> <maxL1Tx> <maxFeeTx> <B> <inflight> <steps> <warmup> <expected.json>
const fs = require("fs");
const path = require("path");
const { tester } = require(path.join(__dirname, "..", "..", "circuits_amd", "node", "index.js"));

async function main() {
    const [file, nTx, L, m1, F, B, inflight, steps, warmup] = process.argv.slice(2, 11).map((x, i) => (i === 0 ? x : Number(x)));
    const expected = process.argv[11] ? JSON.parse(fs.readFileSync(process.argv[11], "utf8")) : null;
    const ctxs = [];
    for (let k = 0; k < inflight; k++) ctxs.push(await tester(`component main = RollupMain(${nTx}, ${L}, ${m1}, ${F});`, { nInstances: B }));
    const each = ctxs[0].packedLayout().bytes;
    const size = fs.statSync(file).size;
    const nDistinct = Math.floor(size / each);
    if (nDistinct < 1) throw new Error("packed file too small");
    // pinned copy of the packed batches, TWICE in a row: any window of B consecutive batches modulo nDistinct is contiguous (one copy per
    // stageRange). As in bench.py's Rotation, context k's instance b holds batch (k * B + b + r) mod nDistinct in round r and every step
    // is a new round: no instance evaluates the batch it held the step before.
    const nWin = Math.max(nDistinct, B);   // (fewer distinct batches than B: the window repeats them)
    const pin = ctxs[0].hostAlloc(each * 2 * nWin);
    const u8 = new Uint8Array(pin);
    const fd = fs.openSync(file, "r");
    for (let j = 0; j < 2 * nWin; j++) fs.readSync(fd, u8, j * each, each, (j % nDistinct) * each);
    fs.closeSync(fd);
    const round = new Array(inflight).fill(0);   // the round staged for context k's next enqueue
    const winOff = (k, r) => (((k * B + r) % nWin) * each);
    const stage = (k) => ctxs[k].stageRange(0, B, pin, winOff(k, round[k]), each);
    // first pass: inputs in, one checked step per context, public outputs against the builder's values
    for (let k = 0; k < inflight; k++) { stage(k); ctxs[k].enqueue(); await ctxs[k].check(true); }
    if (expected) {
        for (let k = 0; k < inflight; k++) {
            const rd = ctxs[k].reader();
            for (const b of [0, B - 1]) {
                const want = expected[((k * B + b) % nWin) % nDistinct];
                if (rd.get(b, "main.hashGlobalInputs").toString() !== want) throw new Error(`hashGlobalInputs mismatch (context ${k}, batch ${b})`);
            }
        }
    }
    // every context is its own chain of step() calls (check previous -> enqueue -> stage next, one work item on the libuv pool each):
    // the contexts in flight advance independently, the JS thread only chains promises
    const mode = process.env.HZ_NODE_MODE || "step";
    const runSync = (n, withStage) => {   // the Python loop verbatim, on the JS thread (HZ_NODE_MODE = sync | nostage: experiments)
        const pending = new Array(inflight).fill(false);
        for (let i = 0; i < n; i++) {
            const k = i % inflight;
            if (pending[k]) ctxs[k].checkSync(true);
            ctxs[k].enqueue();
            if (withStage) { round[k]++; stage(k); }
            pending[k] = true;
        }
        for (let k = 0; k < inflight; k++) if (pending[k]) ctxs[k].checkSync(true);
    };
    const run = async (n) => {
        if (mode === "sync") return runSync(n, true);
        if (mode === "nostage") return runSync(n, false);
        const chains = [];
        for (let k = 0; k < inflight; k++) {
            const mine = Math.floor(n / inflight) + (k < n % inflight ? 1 : 0);
            chains.push((async () => {
                for (let i = 0; i < mine; i++) { round[k]++; await ctxs[k].step(pin, winOff(k, round[k]), 0, B, each, true); }
                if (mine) await ctxs[k].check(true);
            })());
        }
        await Promise.all(chains);
    };
    for (let k = 0; k < inflight; k++) stage(k);
    await run(Math.max(inflight, warmup));
    const t0 = process.hrtime.bigint();
    await run(steps);
    const dt = Number(process.hrtime.bigint() - t0) / 1e9;
    for (let k = 0; k < inflight; k++) { ctxs[k].enqueue(); await ctxs[k].check(true); }   // drain the last staged inputs
    if (expected) {   // the rotation really happened: the last round's batches are what the instances hold
        for (let k = 0; k < inflight; k++) {
            const rd = ctxs[k].reader();
            for (const b of [0, B - 1]) {
                const want = expected[((k * B + b + round[k]) % nWin) % nDistinct];
                if (rd.get(b, "main.hashGlobalInputs").toString() !== want) throw new Error(`hashGlobalInputs mismatch after the timed region (context ${k}, batch ${b}, round ${round[k]})`);
            }
        }
    }
    console.log(JSON.stringify({ value_node: nTx * B * steps / dt, ms_per_step: dt / steps * 1e3, steps, batches_per_launch: B, contexts_in_flight: inflight,
        distinct_batches: Math.min(nDistinct, B * inflight), mode, host: "node " + process.version + " over N-API (circuits_amd/node)", uploads: "inside the timed region (stageRange per step)", rotation: "a new batch per instance and step" }));
}
main().catch((e) => { console.error(e); process.exit(1); });
