import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from circuits_amd import builder as B, lib, ConstraintError
from oracle_binding import OracleCtx
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
hz=lib()
batch=B.synthetic_batch(8, 16, 3, 4, n_accounts=6, exits=2)
inp = dict(batch.get_input())
i = inp["onChain"].index(0)
cases = []
bad = dict(inp); bad["s"] = list(inp["s"]); bad["s"][i] = (inp["s"][i] + 1) % P
cases.append(bad)
bad = dict(inp); bad["imStateRoot"] = list(inp["imStateRoot"]); bad["imStateRoot"][2] = (inp["imStateRoot"][2] + 1) % P
cases.append(bad)
bad = dict(inp); bad["siblings1"] = [list(x) for x in inp["siblings1"]]; bad["siblings1"][i][0] = (bad["siblings1"][i][0] + 1) % P
cases.append(bad)
bad = dict(inp); bad["onChain"] = list(inp["onChain"]); bad["onChain"][0] = 2
cases.append(bad)
for k,bad in enumerate(cases):
    g = hz.ctx("rollup-main", nTx=8, nLevels=16, maxL1Tx=3, maxFeeTx=4)
    o = OracleCtx("rollup-main", 8, 16, 3, 4)
    g.set_inputs(bad); o.set_inputs(bad)
    r = o.run()
    try:
        g.run(); print(k, "gpu: no error; oracle:", r[:4])
    except ConstraintError as e:
        print(k, "gpu:", (e.instance, e.unit, e.constraint_id, e.name), "oracle:", r[:4], (e.lhs,e.rhs)==(r[4],r[5]))
