"use strict";
// Facade test in the style of the reference suites (test/lib/hash-state.test.js:31-57,
// test/rollup-main.test.js:65-72 via test/helpers/helpers.js:147-155), on Node's own assert.
// usage: node run_facade.js <fixture.json> [cpu]
const assert = require("assert");
const fs = require("fs");
const path = require("path");
const { tester, parseMain, deviceCount } = require(path.join(__dirname, "..", "..", "circuits_amd", "node", "index.js"));

async function main() {
    const fx = JSON.parse(fs.readFileSync(process.argv[2], "utf8"));
    const m = parseMain("include \"../src/rollup-main.circom\";\ncomponent main = RollupMain(8, 16, 3, 4);");
    assert.deepStrictEqual(m.params, { nTx: 8, nLevels: 16, maxL1Tx: 3, maxFeeTx: 4 });
    assert.throws(() => parseMain("component main = Sha256(3);"), /not part/);
    if (process.argv[3] === "cpu") {
        if (deviceCount() > 0) { console.log("gpu present, cpu-only checks skipped"); return; }
        await assert.rejects(tester("component main = HashState();"), /no usable gfx950 device/);
        console.log("node facade (cpu): ok");
        return;
    }
    // config 1: hash-state single leaf
    {
        const circuit = await tester("component main = HashState();", { reduceConstraints: false });
        await circuit.loadConstraints();
        const w = await circuit.calculateWitness(fx.hashState.input, { logTrigger: false, logOutput: false, logSet: false });
        assert.strictEqual(w[0], 1n);
        await circuit.assertOut(w, { out: fx.hashState.out });
    }
    // rollup-main batch: assertBatch
    {
        const p = fx.rollupMain.params;
        const circuit = await tester(`component main = RollupMain(${p.nTx}, ${p.nLevels}, ${p.maxL1Tx}, ${p.maxFeeTx});`, { reduceConstraints: false });
        await circuit.loadConstraints();
        assert.strictEqual(circuit.constraints.length, fx.rollupMain.constraints);
        const w = await circuit.calculateWitness(fx.rollupMain.input, { logTrigger: false, logOutput: false, logSet: false });
        await circuit.assertOut(w, { hashGlobalInputs: fx.rollupMain.hashGlobalInputs });
        assert.strictEqual((await circuit.getSignal(w, "main.rollupTx[0].s4.out")).toString(), fx.rollupMain.input.imStateRoot[0]);
        // negative: tampered intermediate root must throw "Constraint doesn't match"
        const bad = JSON.parse(JSON.stringify(fx.rollupMain.input));
        bad.imStateRoot[1] = (BigInt(bad.imStateRoot[1]) + 1n).toString();
        await assert.rejects(circuit.calculateWitness(bad, true), /Constraint doesn't match/);
        // unknown / mis-shaped signals
        await assert.rejects(circuit.calculateWitness(Object.assign({ nope: 1 }, fx.rollupMain.input), true), /Signal not found/);
        const missing = Object.assign({}, fx.rollupMain.input);
        delete missing.oldStateRoot;
        await assert.rejects(circuit.calculateWitness(missing, true), /Not all inputs have been set/);
    }
    // gadget mains, as the reference's unit suites drive them (test/balance-updater.test.js:31-56,170-190;
    // test/lib/decode-float.test.js:28-38; test/fee-accumulator.test.js)
    {
        const circuit = await tester("include \"../src/balance-updater.circom\";\ncomponent main = BalanceUpdater();", { reduceConstraints: false });
        const input = { oldStBalanceSender: 100, oldStBalanceReceiver: 200, amount: 50, loadAmount: 0, feeSelector: 126, onChain: 0, nop: 0, nullifyLoadAmount: 0, nullifyAmount: 0 };
        const w = await circuit.calculateWitness(input, { logOutput: false });
        await circuit.assertOut(w, { newStBalanceSender: 100 - 50 - 5, newStBalanceReceiver: 250, fee2Charge: 5, isP2Nop: 1, isAmountNullified: 0 });
        await assert.rejects(circuit.calculateWitness(Object.assign({}, input, { amount: 98, feeSelector: 200 }), true), /Constraint doesn't match 1 != 0/);
        const df = await tester("component main = DecodeFloat();");
        await df.assertOut(await df.calculateWitness({ in: "0xF8000002FF" }, true), { out: 767n * 10n ** 31n });
        const fa = await tester("component main = FeeAccumulator(4);");
        await fa.assertOut(await fa.calculateWitness({ tokenID: 3, fee2Charge: 7, feePlanTokenID: [1, 3, 3, 4], accFeeIn: [10, 20, 30, 40] }, true), { accFeeOut: [10, 27, 30, 40] });
    }
    // every `component main` of the reference's 16 suites through N-API (inputs from the builder / the suites' literals,
    // expected outputs from the CPU oracle: tests/golden/gen_node_fixture.py)
    let nCases = 0;
    for (const m of fx.mains) {
        const circuit = await tester(m.spec, { reduceConstraints: false });
        for (const c of m.cases) {
            if (c.fail !== undefined) {
                await assert.rejects(circuit.calculateWitness(c.input, true), (e) => e.message.includes("Constraint doesn't match") && e.message.includes(c.fail), `${m.spec} must fail with ${c.fail}`);
                // circom_tester's sanityCheck = false: no assert, the witness comes back as computed
                const w = await circuit.calculateWitness(c.input, { sanityCheck: false });
                assert.strictEqual(w[0], 1n);
            } else {
                const w = await circuit.calculateWitness(c.input, { logOutput: false });
                assert.strictEqual(w[0], 1n);
                await circuit.assertOut(w, c.out);
            }
            nCases++;
        }
    }
    // calculateWitnessBin = the body of a .wtns file; writeWtns writes the same elements
    {
        const circuit = await tester("component main = HashState();");
        const bin = await circuit.calculateWitnessBin(fx.hashState.input, true);
        assert.strictEqual(bin.length, 32 * circuit.nVars);
        const os = require("os");
        const file = path.join(os.tmpdir(), `hz_facade_${process.pid}.wtns`);
        circuit.writeWtns(file, 0);
        const f = fs.readFileSync(file);
        fs.unlinkSync(file);
        assert.strictEqual(f.slice(0, 4).toString(), "wtns");
        assert.ok(f.slice(f.length - bin.length).equals(bin), ".wtns data section != calculateWitnessBin");
        // Poseidon batch through N-API: HashState's digest is Poseidon(4) of its packed inputs
        const { poseidonBatch } = require(path.join(__dirname, "..", "..", "circuits_amd", "node", "index.js"));
        const d = poseidonBatch(5, [fx.hashState.poseidonInputs, fx.hashState.poseidonInputs]);
        assert.strictEqual(d[0].toString(), fx.hashState.out);
        assert.strictEqual(d[1].toString(), fx.hashState.out);
        const dw = poseidonBatch(5, [fx.hashState.poseidonInputs], true);
        assert.strictEqual(dw.witness.length, 96 * (8 * 5 + 60));
    }
    // many instances per run: (a) by signal name, (b) the measured path -- packed inputs in pinned memory, staged upload, enqueue, check
    {
        const p = fx.rollupMain.params;
        const many = fx.rollupMainMany;
        const circuit = await tester(`component main = RollupMain(${p.nTx}, ${p.nLevels}, ${p.maxL1Tx}, ${p.maxFeeTx});`, { nInstances: many.length });
        const r = await circuit.calculateWitnessBatch(many.map((b) => b.input), true);
        many.forEach((b, k) => assert.strictEqual(r.get(k, "main.hashGlobalInputs").toString(), b.hashGlobalInputs));
        const lay = circuit.packedLayout();
        assert.ok(lay.bytes > 0 && lay.inputs.length === Object.keys(many[0].input).length);
        const pin = circuit.hostAlloc(lay.bytes * many.length);
        // reversed order, so that the staged inputs differ from what the context holds
        many.forEach((b, k) => circuit.packInput(many[many.length - 1 - k].input, pin, k * lay.bytes));
        circuit.stageRange(0, many.length, pin, 0, lay.bytes);
        circuit.enqueue();
        await circuit.check(true);
        const rd = circuit.reader();
        many.forEach((b, k) => assert.strictEqual(rd.get(k, "main.hashGlobalInputs").toString(), many[many.length - 1 - k].hashGlobalInputs));
        assert.ok(circuit.devPtr() > 0n);
        // the serving-loop form: step() = check the previous step, enqueue, stage the next inputs, in one hop to the pool
        many.forEach((b, k) => circuit.packInput(b.input, pin, k * lay.bytes));
        await circuit.step(pin, 0, 0, many.length, lay.bytes, true);   // enqueues on the reversed inputs, stages the straight ones
        await circuit.step(null, 0, 0, 0, 0, true);                    // checks that step, enqueues on the straight inputs
        await circuit.check(true);
        many.forEach((b, k) => assert.strictEqual(circuit.reader().get(k, "main.hashGlobalInputs").toString(), b.hashGlobalInputs));
        assert.ok(circuit.witnessTotal() > circuit.nVars);
        // a tampered batch among the staged ones rejects with the instance in the record
        const bad = JSON.parse(JSON.stringify(many[1].input));
        bad.imStateRoot[2] = (BigInt(bad.imStateRoot[2]) + 1n).toString();
        circuit.packInput(bad, pin, 2 * lay.bytes);
        circuit.stageRange(2, 1, pin, 2 * lay.bytes, lay.bytes);
        circuit.enqueue();
        await assert.rejects(circuit.check(true), (e) => /Constraint doesn't match/.test(e.message) && e.constraint.instance === 2);
        await assert.rejects(async () => circuit.packInput(Object.assign({ nope: 1 }, many[0].input), pin, 0), /Signal not found/);
        // failures(): the first violated constraint of EVERY instance of the launch, not only the first one (hz_witness_failures)
        const bad0 = JSON.parse(JSON.stringify(many[0].input));
        bad0.imExitRoot[1] = (BigInt(bad0.imExitRoot[1]) + 1n).toString();
        circuit.packInput(bad0, pin, 0);
        circuit.stageRange(0, 1, pin, 0, lay.bytes);
        circuit.enqueue();
        await assert.rejects(circuit.check(true), (e) => e.constraint.instance === 0);
        const all = await circuit.failures();
        assert.deepStrictEqual(all.map((f) => f.instance), [0, 2]);
        assert.ok(/imExitRoot/.test(all[0].constraintName) && /imStateRoot/.test(all[1].constraintName) && /Constraint doesn't match/.test(all[1].message));
        // a key bit that is not a bit reaches the device as it is and fails the circuit's own boolean constraint (it used to be masked to
        // a bit, so that an input the reference rejects came back as an accepted witness); what a byte cannot carry is refused
        const nb = JSON.parse(JSON.stringify(many[0].input));
        nb.fromBjjCompressed[1][5] = 2;
        circuit.packInput(nb, pin, 0);
        circuit.packInput(many[2].input, pin, 2 * lay.bytes);
        await circuit.step(pin, 0, 0, many.length, lay.bytes, false);    // enqueues on the inputs of the case before (two bad batches), stages the bad bit
        await circuit.step(null, 0, 0, 0, 0, false);                     // that step is rejected: nothing is enqueued behind it ...
        const again = await circuit.failures();                          // ... so the per-instance report of the pipelined loop is still there
        assert.deepStrictEqual(again.map((f) => f.instance), [0, 2]);
        await circuit.step(null, 0, 0, 0, 0, false);                     // and the loop goes on: enqueues on the staged inputs
        await assert.rejects(circuit.check(true), (e) => /fromBjjCompressed boolean/.test(e.message) && e.constraint.instance === 0 && e.constraint.unit === 1 && /2 != 0/.test(e.message));
        nb.fromBjjCompressed[1][5] = 256;
        assert.throws(() => circuit.packInput(nb, pin, 0), RangeError);
        // offsets, strides and counts from JavaScript are validated, not cast
        assert.throws(() => circuit.stageRange(0, 1, pin, -32, lay.bytes), RangeError);
        assert.throws(() => circuit.stageRange(0, 2, pin, 0, Number.MAX_SAFE_INTEGER), RangeError);
        assert.throws(() => circuit.stageRange(0, 1, pin, NaN, lay.bytes), RangeError);
        await assert.rejects(circuit.step(pin, 0, 0, many.length + 1, lay.bytes, false), RangeError);
        circuit.packInput(many[0].input, pin, 0);
        circuit.stageRange(0, 1, pin, 0, lay.bytes);
        circuit.enqueue();
        await circuit.check(true);
    }
    console.log(`node facade: ok (${nCases} cases over ${fx.mains.length} mains + batched path)`);
}
main().catch((e) => { console.error(e); process.exit(1); });
