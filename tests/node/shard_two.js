"use strict";
// One RollupMain batch over TWO Node processes (ranks), the collectives inside the library (hz_shard_step): this file starts itself again
// as rank 1; both ranks set the whole batch's inputs, join the communicator over a Unix socket and make the sharded pass. Rank r then
// holds the signals of its transaction range -- compared with an unsharded circuit of the same process -- and rank 0 the public output.
// usage: node shard_two.js <fixture.json> <socket|rccl> [rank]      (rccl: one rank only -- two ranks cannot share one GPU in RCCL)
const assert = require("assert");
const fs = require("fs");
const os = require("os");
const path = require("path");
const { spawn } = require("child_process");
const { tester, Circuit } = require(path.join(__dirname, "..", "..", "circuits_amd", "node", "index.js"));

async function main() {
    const fx = JSON.parse(fs.readFileSync(process.argv[2], "utf8")).rollupMain;
    const transport = process.argv[3] || "socket";
    const world = transport === "rccl" ? 1 : 2;
    const rank = process.argv[4] ? Number(process.argv[4]) : 0;
    const sock = process.argv[5] || path.join(os.tmpdir(), `hz_shard_${process.pid}.sock`);
    let child = null;
    if (rank === 0 && world > 1) child = spawn(process.execPath, [__filename, process.argv[2], transport, "1", sock], { stdio: "inherit" });
    const p = fx.params;
    const spec = `component main = RollupMain(${p.nTx}, ${p.nLevels}, ${p.maxL1Tx}, ${p.maxFeeTx});`;
    const whole = await tester(spec);
    await whole.calculateWitness(fx.input, true);
    const part = await tester(spec);
    part.joinComm(transport, rank, world, world > 1 ? sock : "");
    for (let pass = 0; pass < 2; pass++) await part.shardStep(pass === 0 ? fx.input : null, true);   // (the second pass re-evaluates the same inputs)
    const [first, count] = Circuit.shardRange(p.nTx, world, rank);
    assert(count > 0 && (world === 1 || count < p.nTx));
    for (let i = first; i < first + count; i++)
        for (const sig of [`main.rollupTx[${i}].s4.out`, `main.rollupTx[${i}].s5.out`])
            assert.strictEqual(part.readSignal(sig), whole.readSignal(sig), `rank ${rank}: ${sig}`);
    if (rank === 0) {
        assert.strictEqual(part.readSignal("main.hashGlobalInputs").toString(), fx.hashGlobalInputs);
        // a constraint violated in the OTHER rank's range is that rank's to report; one in this rank's range rejects here
        const bad = JSON.parse(JSON.stringify(fx.input));
        bad.imStateRoot[first] = (BigInt(bad.imStateRoot[first]) + 1n).toString();
        if (world === 1) await assert.rejects(part.shardStep(bad, true), /Constraint doesn't match/);
    }
    if (child) {
        const code = await new Promise((res) => child.on("exit", res));
        assert.strictEqual(code, 0, "rank 1 failed");
    }
    console.log(`shard_two (${transport}) rank ${rank}: ok`);
}
main().catch((e) => { console.error(e); process.exit(1); });
