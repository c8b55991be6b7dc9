// node wtns_r1cs.js "<Template(params)>" input.json circuit.sym circuit.r1cs out.wtns [bad.r1cs]
// The reference's flow in JavaScript (tools/helpers/actions.js:132-170: witness, then the prover's files): the witness of one input
// written as the .wtns of the COMPILE the .sym / .r1cs belong to, every constraint checked first; with a sixth argument, an .r1cs the
// witness does not satisfy must be refused with no file written.
const fs = require("fs");
const path = require("path");
const assert = require("assert");
const { tester } = require(path.join(__dirname, "..", "..", "circuits_amd", "node"));

(async () => {
    const [spec, inputFile, symFile, r1csFile, out, badFile] = process.argv.slice(2);
    const circuit = await tester(spec);
    const input = JSON.parse(fs.readFileSync(inputFile, "utf8"));
    await circuit.calculateWitness(input, true);
    const sym = fs.readFileSync(symFile, "utf8");
    circuit.writeWtns(out, 0, sym, fs.readFileSync(r1csFile), true);
    assert(fs.existsSync(out));
    assert.throws(() => circuit.writeWtns(out + ".nor1cs", 0, sym), /not stored by this layout/);
    if (badFile) {
        assert.throws(() => circuit.writeWtns(out + ".bad", 0, sym, fs.readFileSync(badFile), true), /constraints of the \.r1cs do not hold/);
        assert(!fs.existsSync(out + ".bad"));
    }
    // the same compile imported ONCE and kept (circuit.importSym): w[] in the compiler's order in memory -- what the reference's
    // calculateWitness hands its callers -- the .wtns from the kept map byte for byte the one above, the constraint check on the exported buffer
    const map = circuit.importSym(sym, fs.readFileSync(r1csFile));
    assert(map.nVars > 1000 && map.solved > 0 && map.derived > 0);
    assert.deepStrictEqual(map.check(0), { bad: 0, first: -1 });
    const w = await map.witness(0);
    assert.strictEqual(w.length, map.nVars);
    assert.strictEqual(w[0], 1n);
    const bin = Buffer.from(await map.witnessBin(0));
    const file = fs.readFileSync(out);
    assert(file.slice(file.length - bin.length).equals(bin), "the .wtns data section is the exported vector");
    map.writeWtns(out + ".map", 0);
    assert(fs.readFileSync(out + ".map").equals(file));
    assert.throws(() => circuit.importSym(sym), /not stored by this layout/);
    if (badFile) {
        const bad = circuit.importSym(sym, fs.readFileSync(badFile));
        const r = bad.check(0);
        assert(r.bad === 1 && r.first >= 0);
        bad.release();
        assert.throws(() => bad.check(0), /released symbol-map handle/);
    }
    // several exports of one circuit pending at once (the export stages through per-context buffers on the context's stream: the work
    // items must not interleave) and release() while they are pending (deferred to the last completion; the handle is dead at once)
    const many = [map.witnessBin(0), map.witnessBin(0), map.witnessBin(0)];
    map.release();
    assert.throws(() => map.witnessBin(0), /released symbol-map handle/);
    assert.throws(() => map.check(0), /released symbol-map handle/);
    for (const ab of await Promise.all(many)) assert(Buffer.from(ab).equals(bin), "concurrent exports of one circuit return the same vector");
    console.log("wtns_r1cs: ok");
})().catch((e) => { console.error(e); process.exit(1); });
