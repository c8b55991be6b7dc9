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
    console.log("wtns_r1cs: ok");
})().catch((e) => { console.error(e); process.exit(1); });
