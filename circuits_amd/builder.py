"""Batch builder: builds circuit INPUTS for RollupMain / RollupTx / Withdraw from a sequence of
transactions -- the role `@hermeznetwork/commonjs` (RollupDB.buildBatch / BatchBuilder.addTx /
build / getInput / getHashInputs, called by the reference at test/helpers/helpers.js:46,148 and
tools/generate-input.js:70-107) plays for the reference. That package is not on disk; this is a
from-scratch restatement of the protocol rules the circuits themselves enforce (src/*.circom),
plus the sparse-Merkle-tree update rules of circomlib's SMT (Poseidon hashes, key bits LSB first).

Caller-side code: it never computes a witness. Field/curve arithmetic comes from
circuits_amd/libhz_host.so (the product's own headers compiled for the host).
"""
import ctypes
import hashlib
import json
import os

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
SUBORDER = 2736030358979909402780800718157159386076813972158567259200215660948447373041
BASE8 = (5299619240641551281634865583518297030282874472190772894086521144482721001553,
         16950150798460657717958625567821834550301663161624707787222815936182638968203)
CONST_SIG = 3322668559
EXIT_IDX = 1
_HERE = os.path.dirname(os.path.abspath(__file__))


class Host:
    """ctypes view of libhz_host.so."""

    def __init__(self):
        path = os.path.join(_HERE, "libhz_host.so")
        if not os.path.exists(path):
            raise RuntimeError("%s missing: run __graft_entry__.build()" % path)
        self.c = ctypes.CDLL(path)
        self._o = ctypes.create_string_buffer(32)
        self._ox = ctypes.create_string_buffer(32)
        self._oy = ctypes.create_string_buffer(32)

    def poseidon(self, xs):
        buf = b"".join(int(x % P).to_bytes(32, "little") for x in xs)
        self.c.hzb_poseidon(len(xs), buf, self._o)
        return int.from_bytes(self._o.raw, "little")

    def poseidon_many(self, t, n, data):
        """n hashes of t - 1 inputs each from a bytes-like of canonical 32-byte elements -> bytes of n digests (one call)"""
        out = ctypes.create_string_buffer(32 * max(n, 1))
        self.c.hzb_poseidon_many(ctypes.c_int(t - 1), ctypes.c_uint64(n), bytes(data), out)
        return out.raw[:32 * n]

    def bjj_mul(self, pt, k):
        b = lambda v: int(v).to_bytes(32, "little")  # noqa: E731
        self.c.hzb_bjj_mul(b(pt[0]), b(pt[1]), b(k % (1 << 256)), self._ox, self._oy)
        return int.from_bytes(self._ox.raw, "little"), int.from_bytes(self._oy.raw, "little")


_host = None


def host():
    global _host
    if _host is None:
        _host = Host()
    return _host


# ---- fee table / float40 (reference src/compute-fee.circom:105-109, src/lib/decode-float.circom) ----
_FEE = None


def fee_table():
    global _FEE
    if _FEE is None:
        _FEE = [int(x) for x in json.load(open(os.path.join(_HERE, "fee_table.json")))["table"]]
    return _FEE


def compute_fee(amount, sel):
    t = fee_table()[sel]
    return (amount * t) >> 60 if sel < 192 else amount * t


def float2fix(f):
    return (f & ((1 << 35) - 1)) * 10 ** (f >> 35)


def floor_fix2float(v):
    """largest float40 not above v"""
    if v == 0:
        return 0
    best = 0
    for e in range(32):
        m = v // 10 ** e
        if m < (1 << 35):
            cand = m + (e << 35)
            if float2fix(cand) > float2fix(best):
                best = cand
            break
    return best


def fix2float(v):
    f = floor_fix2float(v)
    if float2fix(f) != v:
        raise ValueError("amount %d is not float40-representable" % v)
    return f


# ---- circomlib-compatible sparse Merkle tree ---------------------------------------------------
class DagHasher:
    """Deferred Poseidon for the device-side batch builder. poseidon() records a job and returns a REFERENCE
    (a negative integer: field values are never negative) that can be stored in the tree, used as a dictionary key and fed to
    later jobs; no hash is computed while the batch is walked. Every node a job depends on lies deeper in a tree (own child:
    same update; sibling: an earlier update), so the dependency depth of the whole batch is nLevels + 3 whatever the number of
    transactions: resolve() sorts the jobs by depth and evaluates each depth as ONE launch of hz_poseidon_dag (all node
    versions of a tree level in parallel) instead of 2*(nLevels+1) dependent hashes per transaction on a CPU thread
    (reference: @hermeznetwork/commonjs BatchBuilder via test/helpers/helpers.js:46, tools/generate-input.js:70-107)."""
    CONST = 1 << 31
    _PAD = [0] * 6

    def __init__(self, evaluate):
        self.evaluate = evaluate   # (vals: bytearray, job_in, job_out, seg_t, seg_first, seg_count) -> device ms or None
        self.stats = {"jobs": 0, "segments": 0, "device_ms": 0.0, "resolve_s": 0.0}
        self._reset()

    def _reset(self):
        import array
        self.n = 0
        self.wave, self.arity = [], []
        self.flat = array.array("I")
        self.consts, self.cidx = [], {}
        self.out = None

    def poseidon(self, xs):
        w, idx = 0, []
        for x in xs:
            if x < 0:
                j = -x - 1
                idx.append(j)
                if self.wave[j] >= w:
                    w = self.wave[j] + 1
            else:
                c = self.cidx.get(x)
                if c is None:
                    c = self.cidx[x] = len(self.consts)
                    self.consts.append(x)
                idx.append(self.CONST | c)
        self.wave.append(w)
        self.arity.append(len(idx))
        self.flat.extend(idx + self._PAD[len(idx):])
        self.n += 1
        return -self.n

    def resolve(self):
        """Evaluates the recorded jobs; returns value(ref_or_int) -> int and forgets the jobs."""
        import time
        import numpy as np
        n = self.n
        if n == 0:
            return lambda x: x
        t0 = time.time()
        wave = np.asarray(self.wave, dtype=np.int64)
        ar = np.asarray(self.arity, dtype=np.int64)
        order = np.lexsort((ar, wave))
        a = np.frombuffer(self.flat, dtype=np.uint32).reshape(n, 6).astype(np.int64)
        a = np.where(a >= self.CONST, (a & (self.CONST - 1)) + n, a)
        job_in = np.ascontiguousarray(a[order].astype(np.uint32))
        job_out = np.ascontiguousarray(order.astype(np.uint32))
        key = wave[order] * 8 + ar[order]
        cut = np.flatnonzero(np.diff(key)) + 1
        seg_first = np.concatenate(([0], cut)).astype(np.uint64)
        seg_count = np.diff(np.concatenate((seg_first, [n]))).astype(np.uint64)
        seg_t = (ar[order][seg_first.astype(np.int64)] + 1).astype(np.uint32)
        vals = bytearray(32 * n) + b"".join(int(c % P).to_bytes(32, "little") for c in self.consts)
        ms = self.evaluate(vals, job_in, job_out, seg_t, seg_first, seg_count)
        out = [int.from_bytes(vals[32 * i:32 * i + 32], "little") for i in range(n)]
        self.stats["jobs"] += n
        self.stats["segments"] += len(seg_t)
        self.stats["device_ms"] += ms or 0.0
        self.stats["resolve_s"] += time.time() - t0
        self._reset()
        return lambda x: x if x >= 0 else out[-x - 1]


class DenseState:
    """A pre-populated state of N = 2^k accounts (idx first_idx .. first_idx + N - 1) whose tree is held as arrays instead of a
    dictionary of nodes: with consecutive keys every residue class modulo 2^k holds exactly one key, so the circomlib tree (key bits
    LSB first, a leaf at the shallowest level where it is alone) is the perfect binary tree of depth k -- node (d, p) covers the
    keys = p (mod 2^d), its children are (d + 1, p) and (d + 1, p + 2^d), the leaves sit at depth k. Built bottom-up, one batched
    Poseidon call per level (`hash_rows(t, n, bytes) -> bytes`: the device through hz_poseidon_batch, or the host one hash at a
    time), 3 N hashes in all; what bench.py's `deep_state` line uses for 2^20 accounts (the proofs then hash ~21 levels instead of
    ~14). The accounts follow synthetic_batch's recipe (token 1, one of `n_keys` keys, float40 balance below 2^95), drawn with numpy
    from `seed`. The object is read-only: updates live in the SMT's own node dictionary / RollupDB.leaves on top of it."""

    def __init__(self, k, first_idx, seed, n_keys, key_idx, mant, expo, levels, value):
        self.k, self.first_idx, self.seed, self.n_keys = k, first_idx, seed, n_keys
        self.N = 1 << k
        self.key_idx, self.mant, self.expo, self.levels, self.value = key_idx, mant, expo, levels, value
        self._keys = None

    @property
    def root(self):
        return int.from_bytes(self.levels[0][0].tobytes(), "little")

    def keys(self):
        if self._keys is None:
            self._keys = [Account(self.seed * 1000 + i) for i in range(self.n_keys)]
        return self._keys

    def key_of(self, p):
        """the key (account index) whose residue modulo 2^k is p"""
        r = (p - self.first_idx) % self.N
        return self.first_idx + r

    def has(self, idx):
        return self.first_idx <= idx < self.first_idx + self.N

    def state(self, idx):
        j = idx - self.first_idx
        a = self.keys()[int(self.key_idx[j])]
        return {"tokenID": 1, "nonce": 0, "sign": a.sign, "balance": int(self.mant[j]) * 10 ** int(self.expo[j]), "ay": a.ay, "ethAddr": a.eth_addr}

    def hash_at(self, depth, prefix):
        return int.from_bytes(self.levels[depth][prefix].tobytes(), "little")

    def node(self, depth, prefix):
        """the base tree's node at (depth, prefix) in SMT.nodes form"""
        if depth == self.k:
            return ("leaf", self.key_of(prefix), int.from_bytes(self.value[(self.key_of(prefix) - self.first_idx)].tobytes(), "little"))
        lv = self.levels[depth + 1]
        return ("mid", int.from_bytes(lv[prefix].tobytes(), "little"), int.from_bytes(lv[prefix + (1 << depth)].tobytes(), "little"))

    @staticmethod
    def build(k, seed=0x48455A31, first_idx=256, n_keys=8, hash_rows=None):
        import numpy as np
        N = 1 << k
        if hash_rows is None:
            hash_rows = host().poseidon_many   # the host library, one call per level (a million accounts want the device)
        rng = np.random.default_rng(seed)
        keys = [Account(seed * 1000 + i) for i in range(n_keys)]
        key_idx = rng.integers(0, n_keys, size=N, dtype=np.uint8)
        mant = rng.integers(1, 1 << 35, size=N, dtype=np.uint64)
        expo = rng.integers(0, 19, size=N, dtype=np.uint8)

        def col(vals):   # [N] python ints -> [N, 32] uint8 little-endian
            return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(-1, 32)
        e0_k = col([1 + (a.sign << 72) for a in keys])
        ay_k = col([a.ay for a in keys])
        eth_k = col([a.eth_addr for a in keys])
        bal = col([int(m) * 10 ** int(e) for m, e in zip(mant.tolist(), expo.tolist())])
        rows = np.stack([e0_k[key_idx], bal, ay_k[key_idx], eth_k[key_idx]], axis=1)            # [N, 4, 32]
        value = np.frombuffer(hash_rows(5, N, rows.tobytes()), dtype=np.uint8).reshape(N, 32)   # state hash of account j
        # leaves at depth k, indexed by residue p: the account j = (p - first_idx) mod N
        p = np.arange(N, dtype=np.int64)
        j = (p - first_idx) % N
        keycol = np.zeros((N, 32), dtype=np.uint8)
        kk = (first_idx + j).astype(np.uint64)
        for b in range(8):
            keycol[:, b] = ((kk >> np.uint64(8 * b)) & np.uint64(0xFF)).astype(np.uint8)
        one = np.zeros((N, 32), dtype=np.uint8)
        one[:, 0] = 1
        rows = np.stack([keycol, value[j], one], axis=1)
        levels = [None] * (k + 1)
        levels[k] = np.frombuffer(hash_rows(4, N, rows.tobytes()), dtype=np.uint8).reshape(N, 32)
        for d in range(k - 1, -1, -1):
            n = 1 << d
            rows = np.stack([levels[d + 1][:n], levels[d + 1][n:2 * n]], axis=1)
            levels[d] = np.frombuffer(hash_rows(3, n, rows.tobytes()), dtype=np.uint8).reshape(n, 32)
        return DenseState(k, first_idx, seed, n_keys, key_idx, mant, expo, levels, value)

    def save(self, path):
        import numpy as np
        arrs = {"meta": np.array([self.k, self.first_idx, self.seed, self.n_keys], dtype=np.int64), "key_idx": self.key_idx, "mant": self.mant,
                "expo": self.expo, "value": self.value}
        for d, lv in enumerate(self.levels):
            arrs["lv%d" % d] = lv
        np.savez(path, **arrs)

    @staticmethod
    def load(path):
        import numpy as np
        z = np.load(path)
        k, first_idx, seed, n_keys = (int(x) for x in z["meta"])
        return DenseState(k, first_idx, seed, n_keys, z["key_idx"], z["mant"], z["expo"], [z["lv%d" % d] for d in range(k + 1)], z["value"])


class _BaseLeaves(dict):
    """RollupDB.leaves over a DenseState: accounts that have not been written yet come from the base"""

    def __init__(self, base):
        super().__init__()
        self.base = base

    def __missing__(self, idx):
        if not self.base.has(idx):
            raise KeyError(idx)
        return self.base.state(idx)

    def __contains__(self, idx):
        return dict.__contains__(self, idx) or self.base.has(idx)


class SMT:
    def __init__(self, hasher=None, base=None):
        self.h = hasher or host()
        self.lazy = isinstance(self.h, DagHasher)
        self.fresh = []   # lazy mode: node ids created since the last rekey()
        self.base = base  # DenseState: the nodes nobody has rewritten yet (found by position, not by hash)
        self.root = base.root if base is not None else 0
        self.nodes = {}  # hash -> ("leaf", key, value) | ("mid", left, right)

    def _put(self, k, node):
        self.nodes[k] = node
        if self.lazy:
            self.fresh.append(k)

    def rekey(self, val):
        """lazy mode, after DagHasher.resolve(): node ids (references) become the hashes they stand for"""
        for k in self.fresh:
            n = self.nodes.pop(k, None)
            if n is not None:
                self.nodes[val(k)] = (n[0], n[1], val(n[2])) if n[0] == "leaf" else (n[0], val(n[1]), val(n[2]))
        self.fresh = []
        self.root = val(self.root)

    def _hash1(self, k, v):
        return self.h.poseidon([k, v, 1])

    def _hash0(self, l, r):
        return self.h.poseidon([l, r])

    def find(self, key):
        node, sib, lvl = self.root, [], 0
        while True:
            if node == 0:
                return {"found": False, "siblings": sib, "notFoundKey": key, "notFoundValue": 0, "isOld0": True}
            n = self.nodes.get(node)
            if n is None:   # a node of the pre-populated tree that no update has replaced: known by its position on the path
                if self.base is None:
                    raise KeyError(node)
                # below depth k only a base LEAF can appear (pushed down by the insertion of a key of the same residue class)
                d = min(lvl, self.base.k)
                if self.base.hash_at(d, key & ((1 << d) - 1)) != node:
                    raise KeyError(node)
                n = self.base.node(d, key & ((1 << d) - 1))
            if n[0] == "leaf":
                if n[1] == key:
                    return {"found": True, "siblings": sib, "foundValue": n[2], "isOld0": False}
                return {"found": False, "siblings": sib, "notFoundKey": n[1], "notFoundValue": n[2], "isOld0": False, "node": node}
            if (key >> lvl) & 1:
                sib.append(n[1])
                node = n[2]
            else:
                sib.append(n[2])
                node = n[1]
            lvl += 1

    def _up(self, key, leaf_hash, sib):
        rt = leaf_hash
        for i in range(len(sib) - 1, -1, -1):
            l, r = (sib[i], rt) if (key >> i) & 1 else (rt, sib[i])
            rt = self._hash0(l, r)
            self._put(rt, ("mid", l, r))
        return rt

    def insert(self, key, value):
        f = self.find(key)
        if f["found"]:
            raise KeyError("key exists")
        res = {"oldRoot": self.root, "isOld0": f["isOld0"], "oldKey": f["notFoundKey"], "oldValue": f["notFoundValue"]}
        sib = list(f["siblings"])
        full = list(sib)
        if not f["isOld0"]:
            ok = f["notFoundKey"]
            i = len(full)
            while ((ok >> i) & 1) == ((key >> i) & 1):
                full.append(0)
                i += 1
            full.append(f["node"])   # the leaf found on the way = hash1(oldKey, oldValue)
        lh = self._hash1(key, value)
        self._put(lh, ("leaf", key, value))
        self.root = self._up(key, lh, full)
        if not f["isOld0"]:
            full.pop()
        while full and full[-1] == 0:
            full.pop()
        res["siblings"] = full
        res["newRoot"] = self.root
        return res

    def update(self, key, value):
        f = self.find(key)
        if not f["found"]:
            raise KeyError("key not found")
        res = {"oldRoot": self.root, "oldKey": key, "oldValue": f["foundValue"], "siblings": list(f["siblings"])}
        lh = self._hash1(key, value)
        self._put(lh, ("leaf", key, value))
        self.root = self._up(key, lh, f["siblings"])
        res["newRoot"] = self.root
        return res


# ---- accounts / signatures ------------------------------------------------------------------------
class Account:
    """Synthetic Hermez account: BabyJubjub key k (A = k*Base8), 160-bit address."""

    def __init__(self, seed):
        d = hashlib.sha512(b"hz-account-%d" % seed).digest()
        self.k = int.from_bytes(d[:32], "little") % SUBORDER or 1
        self.eth_addr = int.from_bytes(d[32:52], "little")
        self.ax, self.ay = host().bjj_mul(BASE8, self.k)
        self.sign = 1 if self.ax > (P - 1) // 2 else 0
        self.bjj_compressed = self.ay | (self.sign << 255)

    def sign_msg(self, msg):
        """EdDSA-Poseidon: S*B8 == R8 + 8*H(R8,A,M)*A (circomlib eddsaposeidon.circom)."""
        d = hashlib.sha512(self.k.to_bytes(32, "little") + int(msg).to_bytes(32, "little")).digest()
        r = int.from_bytes(d, "little") % SUBORDER or 1
        r8 = host().bjj_mul(BASE8, r)
        hm = host().poseidon([r8[0], r8[1], self.ax, self.ay, msg])
        s = (r + 8 * hm * self.k) % SUBORDER
        return {"r8x": r8[0], "r8y": r8[1], "s": s}


_SHA_K = None


def sha256_bits(bits):
    """SHA-256 of a bit string (FIPS 180-4 padding at bit granularity; the circuit's Sha256(nBits) hashes bit strings whose length
    need not be a multiple of 8, e.g. nLevels = 10). Byte-aligned inputs go through hashlib."""
    global _SHA_K
    if len(bits) % 8 == 0:
        return hashlib.sha256(bytes(sum(bits[8 * i + k] << (7 - k) for k in range(8)) for i in range(len(bits) // 8))).digest()
    if _SHA_K is None:
        pr = [n for n in range(2, 312) if all(n % d for d in range(2, int(n ** 0.5) + 1))][:64]

        def frac_root(n, k):
            # floor(2^32 * frac(n^(1/k))) by integer root of n * 2^(32k)
            x, lo, hi = n << (32 * k), 0, 1 << 40
            while lo < hi:
                mid = (lo + hi + 1) // 2
                if mid ** k <= x:
                    lo = mid
                else:
                    hi = mid - 1
            return lo & 0xFFFFFFFF
        _SHA_K = ([frac_root(q, 3) for q in pr], [frac_root(q, 2) for q in pr[:8]])
    K, H0 = _SHA_K
    msg = list(bits) + [1]
    msg += [0] * ((448 - len(msg)) % 512)
    msg += [(len(bits) >> (63 - k)) & 1 for k in range(64)]
    h = list(H0)
    rotr = lambda x, r: ((x >> r) | (x << (32 - r))) & 0xFFFFFFFF   # noqa: E731
    for b in range(0, len(msg), 512):
        w = [sum(msg[b + 32 * i + k] << (31 - k) for k in range(32)) for i in range(16)]
        for i in range(16, 64):
            s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3)
            s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10)
            w.append((w[i - 16] + s0 + w[i - 7] + s1) & 0xFFFFFFFF)
        a, bb, c, d, e, f, g, hh = h
        for i in range(64):
            t1 = (hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i]) & 0xFFFFFFFF
            t2 = ((rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & bb) ^ (a & c) ^ (bb & c))) & 0xFFFFFFFF
            a, bb, c, d, e, f, g, hh = (t1 + t2) & 0xFFFFFFFF, a, bb, c, (d + t1) & 0xFFFFFFFF, e, f, g
        h = [(x + y) & 0xFFFFFFFF for x, y in zip(h, (a, bb, c, d, e, f, g, hh))]
    return b"".join(x.to_bytes(4, "big") for x in h)


def hash_state(st, hasher=None):
    e0 = st["tokenID"] + (st["nonce"] << 32) + (st["sign"] << 72)
    return (hasher or host()).poseidon([e0, st["balance"], st["ay"], st["ethAddr"]])


def build_tx_compressed_data(tx, chain_id):
    return (CONST_SIG | (chain_id << 32) | (tx.get("fromIdx", 0) << 48) | (tx.get("toIdx", 0) << 96) | (tx.get("tokenID", 0) << 144) |
            (tx.get("nonce", 0) << 176) | (tx.get("userFee", 0) << 216) | (tx.get("toBjjSign", 0) << 224))


def build_tx_compressed_data_v2(tx):
    return (tx.get("fromIdx", 0) | (tx.get("toIdx", 0) << 48) | (tx.get("amountF", 0) << 96) | (tx.get("tokenID", 0) << 136) |
            (tx.get("nonce", 0) << 168) | (tx.get("userFee", 0) << 208) | (tx.get("toBjjSign", 0) << 216))


def build_hash_sig(tx, chain_id):
    e1 = tx.get("toEthAddr", 0) | (tx.get("amountF", 0) << 160) | (tx.get("maxNumBatch", 0) << 200)
    return host().poseidon([build_tx_compressed_data(tx, chain_id), e1, tx.get("toBjjAy", 0), tx.get("rqTxCompressedDataV2", 0),
                            tx.get("rqToEthAddr", 0), tx.get("rqToBjjAy", 0)])


def _device_dag_evaluator(device):
    from . import lib
    L = lib()
    if L.device_count() <= 0:
        raise RuntimeError("RollupDB(device=%d): no usable gfx950 device (the device batch builder has no CPU fallback)" % device)
    return lambda *a: L.poseidon_dag(*a, device=device)


class RollupDB:
    def __init__(self, chain_id=1, device=None, dag_evaluator=None, first_idx=256, base=None):
        """device=N: the Merkle / state hashing of every batch runs on GPU N (DagHasher); default: host hashing, one at a time.
        Both produce identical trees and circuit inputs. first_idx: index of the first account created (the protocol reserves
        0..255; a tree of nLevels = 8 -- BASELINE config 2 -- only has room below 256, the circuit itself has no such constant)."""
        self.chain_id = chain_id
        self.hasher = host()
        if device is not None or dag_evaluator is not None:
            self.hasher = DagHasher(dag_evaluator or _device_dag_evaluator(device))
        self.lazy = isinstance(self.hasher, DagHasher)
        self.state = SMT(self.hasher, base)
        self.leaves = {} if base is None else _BaseLeaves(base)  # idx -> state dict
        self.last_idx = first_idx - 1 if base is None else base.first_idx + base.N - 1
        self.num_batch = 0
        self.exit_trees = {}

    def hash_state(self, st):
        return hash_state(st, self.hasher)

    def flush(self, extra_trees=()):
        """lazy mode: evaluates the pending hashes on the device and turns the references held by the trees into values;
        returns value(ref_or_int) for whatever else holds references"""
        if not self.lazy:
            return lambda x: x
        val = self.hasher.resolve()
        for t in (self.state,) + tuple(extra_trees):
            t.rekey(val)
        return val

    def build_batch(self, n_tx, n_levels, max_l1, max_fee):
        return BatchBuilder(self, n_tx, n_levels, max_l1, max_fee)


class BatchBuilder:
    def __init__(self, db, n_tx, n_levels, max_l1, max_fee):
        self.db, self.nTx, self.L, self.maxL1, self.F = db, n_tx, n_levels, max_l1, max_fee
        self.txs, self.fee_tokens, self.fee_idxs = [], [], []
        self.current_num_batch = db.num_batch + 1
        self.built = False

    def add_tx(self, tx):
        if len(self.txs) >= self.nTx:
            raise ValueError("batch full")
        self.txs.append(dict(tx))

    def add_token(self, token_id):
        self.fee_tokens.append(token_id)

    def add_fee_idx(self, idx):
        self.fee_idxs.append(idx)

    # -- helpers
    def _pad(self, sib):
        return list(sib) + [0] * (self.L + 1 - len(sib))

    def build(self):
        db, L, F, nTx = self.db, self.L, self.F, self.nTx
        n_l1 = sum(1 for t in self.txs if t.get("onChain"))
        if n_l1 > self.maxL1:
            raise ValueError("too many L1 txs")
        inp = {k: [] for k in (
            "txCompressedData amountF txCompressedDataV2 fromIdx auxFromIdx toIdx auxToIdx toBjjAy toEthAddr maxNumBatch onChain newAccount "
            "rqOffset rqTxCompressedDataV2 rqToEthAddr rqToBjjAy s r8x r8y loadAmountF fromEthAddr fromBjjCompressed tokenID1 nonce1 sign1 "
            "balance1 ay1 ethAddr1 siblings1 isOld0_1 oldKey1 oldValue1 tokenID2 nonce2 sign2 balance2 ay2 ethAddr2 siblings2 newExit isOld0_2 "
            "oldKey2 oldValue2 imOnChain imOutIdx imStateRoot imExitRoot imAccFeeOut").split()}
        db.flush()   # hashes queued outside a batch (direct state construction)
        inp["oldLastIdx"] = db.last_idx
        inp["oldStateRoot"] = db.state.root
        inp["globalChainID"] = db.chain_id
        inp["currentNumBatch"] = self.current_num_batch
        plan = list(self.fee_tokens) + [0] * (F - len(self.fee_tokens))
        inp["feePlanTokens"] = plan
        acc_fee = [0] * F
        exit_tree = SMT(db.hasher)
        exit_leaves = {}
        self.tx_meta = []  # per-tx facts used by get_single_tx_input
        ordered = [t for t in self.txs if t.get("onChain")] + [t for t in self.txs if not t.get("onChain")]
        self.txs = ordered
        ETH_ADDR_ANY = (1 << 160) - 1
        # Atomic transactions (src/rq-tx-verifier.circom:34-45, src/rollup-main.circom:286-309): a tx with rqOffset k signs
        # the txCompressedDataV2 / toEthAddr / toBjjAy of the tx at i+1..i+3 (k = 1..3) or i-4..i-1 (k = 4..7). Those fields
        # hold the neighbour's nonce, so nonces are assigned in a pre-pass (each L2 tx consumes one of its sender's).
        nonce_sim = {}
        for t in ordered:
            if not t.get("onChain") and t.get("fromIdx"):
                f = t["fromIdx"]
                cur = nonce_sim.get(f, db.leaves[f]["nonce"] if f in db.leaves else 0)
                t.setdefault("nonce", cur)
                nonce_sim[f] = cur + 1
            if "amountF" not in t or "amount" in t:
                t["amountF"] = fix2float(t.get("amount", 0))
        for i, t in enumerate(ordered):
            k = t.get("rqOffset", 0)
            if k and "rqTxCompressedDataV2" not in t:   # explicit rq fields (what the signer committed to) are kept as given
                j = i + k if k <= 3 else i - (8 - k)
                if not (0 <= j < len(ordered)) or ordered[j].get("onChain"):
                    raise ValueError("rqOffset %d of tx %d does not point at an L2 tx of this batch" % (k, i))
                t["rqTxCompressedDataV2"] = build_tx_compressed_data_v2(ordered[j])
                t["rqToEthAddr"] = ordered[j].get("toEthAddr", 0)
                t["rqToBjjAy"] = ordered[j].get("toBjjAy", 0)
        for i in range(nTx):
            tx = ordered[i] if i < len(ordered) else {"onChain": 0, "nop": True}
            # State transition of one transaction, restated from the circuit itself: selectors of
            # src/rollup-tx-states.circom:99-313, balances of src/balance-updater.circom:56-105, leaf
            # multiplexers and processors of src/rollup-tx.circom:318-591 (the JS BatchBuilder the
            # reference calls at test/helpers/helpers.js:46 is not on disk).
            on = 1 if tx.get("onChain") else 0
            from_idx, to_idx = tx.get("fromIdx", 0), tx.get("toIdx", 0)
            amount = tx.get("amount", 0)
            amount_f = tx["amountF"] if "amountF" in tx and "amount" not in tx else fix2float(amount)
            amount = float2fix(amount_f)
            tx["amountF"] = amount_f
            load_f = tx.get("loadAmountF", 0)
            load_amount = float2fix(load_f)
            token = tx.get("tokenID", 0)
            user_fee = tx.get("userFee", 0)
            new_account = 1 if (on and from_idx == 0) else 0
            aux_from = 0
            aux_to = tx.get("auxToIdx", 0)
            if not on and to_idx == 0 and from_idx and "auxToIdx" not in tx:
                # transfer to ethAddr / to Bjj: the coordinator looks the receiver up (lowest idx with that address -- and key, when the
                # address is the "any" address -- holding the token), as the reference's JS BatchBuilder does for its suites
                to_eth = tx.get("toEthAddr", 0)
                for cand in sorted(db.leaves):
                    lf = db.leaves[cand]
                    if lf["tokenID"] != tx.get("tokenID", 0):
                        continue
                    if to_eth != (1 << 160) - 1 and lf["ethAddr"] == to_eth:
                        aux_to = cand
                        break
                    if to_eth == (1 << 160) - 1 and lf["ay"] == tx.get("toBjjAy", 0) and lf["sign"] == tx.get("toBjjSign", 0):
                        aux_to = cand
                        break
            zero_state = {"tokenID": 0, "nonce": 0, "sign": 0, "balance": 0, "ay": 0, "ethAddr": 0}
            st1, st2 = dict(zero_state), dict(zero_state)
            sib1, sib2 = [], []
            isold1 = isold2 = oldk1 = oldk2 = oldv1 = oldv2 = 0
            new_exit = 0
            sig = {"r8x": 0, "r8y": 0, "s": 0}
            bjj = tx.get("fromBjjCompressed", 0)
            from_eth = tx.get("fromEthAddr", 0)
            if not on and (load_f or new_account):
                raise ValueError("loadAmount / newAccount on an L2 tx (the circuit rejects this tx)")
            if new_account:
                db.last_idx += 1
                aux_from = db.last_idx
            final_from = aux_from if new_account else from_idx
            final_to = aux_to if (not on and to_idx == 0) else to_idx
            is_exit = final_to == EXIT_IDX
            nop = final_from == 0
            is_nullified = 0
            if not nop:
                # ---- sender leaf as processor 1 sees it
                if new_account:
                    ay = (bjj & ((1 << 254) - 1)) % P   # a field element in the circuit (Bits2Num of 254 bits): an invalid key may exceed r
                    sg = (bjj >> 255) & 1
                    # coordinator-chosen leaf data for an INSERT: the circuit takes every field from the tx
                    old1 = {"tokenID": token, "nonce": 0, "sign": sg, "balance": 0, "ay": ay, "ethAddr": from_eth}
                    st1 = dict(old1)
                else:
                    if from_idx not in db.leaves:
                        raise ValueError("sender account %d does not exist" % from_idx)
                    old1 = dict(db.leaves[from_idx])
                    st1 = dict(old1)
                if not on and token != old1["tokenID"]:
                    raise ValueError("L2 tokenID does not match the sender leaf (the circuit rejects this tx)")
                # ---- processor 2 key (src/rollup-tx-states.circom:213-221)
                key2 = final_from if is_exit else (final_to if amount else 0)
                p2_insert = is_exit and key2 not in exit_leaves
                # ---- nullifiers (L1 only)
                not_create = on and not new_account
                null_eth = bool(not_create and amount and from_eth != old1["ethAddr"])
                null_tok1 = bool(not_create and token != old1["tokenID"])
                null_load = null_tok1 and load_amount != 0
                # ---- balances
                fee = compute_fee(amount, user_fee) if not on else 0
                eff_load = 0 if (null_load or not on) else load_amount

                # the leaf processor 2 works on; a state-tree receiver is read after processor 1 has written the sender
                def receiver_leaf(sender_new):
                    if is_exit:
                        return dict(exit_leaves[key2]) if key2 in exit_leaves else None
                    if key2 == final_from:
                        return dict(sender_new)
                    if key2 not in db.leaves:
                        raise ValueError("receiver account %d does not exist" % key2)
                    return dict(db.leaves[key2])
                # tokenID2 does not depend on balances: probe the leaf before processor 1 runs
                probe = receiver_leaf(old1) if amount else None
                tok2 = probe["tokenID"] if probe is not None else None
                null_tok2 = bool(on and amount and not p2_insert and probe is not None and token != tok2)
                null_amount = bool(null_eth or null_tok2 or (null_tok1 and amount != 0))
                eff_amount2 = 0 if null_amount else amount
                underflow_ok = old1["balance"] + eff_load - eff_amount2 - fee >= 0
                if not on and not underflow_ok:
                    raise ValueError("L2 underflow (the circuit rejects this tx)")
                eff_amount3 = eff_amount2 if underflow_ok else 0
                is_nullified = 0 if (not null_amount and underflow_ok) else 1
                new1 = dict(old1)
                new1["balance"] = old1["balance"] + eff_load - eff_amount3 - fee
                new1["nonce"] = old1["nonce"] + (0 if on else 1)
                if not on:
                    tx.setdefault("nonce", old1["nonce"])
                if new_account:
                    res = db.state.insert(final_from, db.hash_state(new1))
                    isold1 = 1 if res["isOld0"] else 0
                    oldk1 = 0 if res["isOld0"] else res["oldKey"]
                    oldv1 = 0 if res["isOld0"] else res["oldValue"]
                else:
                    res = db.state.update(final_from, db.hash_state(new1))
                db.leaves[final_from] = new1
                sib1 = res["siblings"]
                if not on and token in plan:
                    acc_fee[plan.index(token)] += fee
                # ---- processor 2: NOP unless the transaction carries an amount (nullified or not)
                if amount:
                    if is_exit:
                        if p2_insert:
                            new_exit = 1
                            enew = {"tokenID": old1["tokenID"], "nonce": 0, "sign": old1["sign"], "balance": eff_amount3, "ay": old1["ay"], "ethAddr": old1["ethAddr"]}
                            r2 = exit_tree.insert(key2, db.hash_state(enew))
                            isold2 = 1 if r2["isOld0"] else 0
                            oldk2 = 0 if r2["isOld0"] else r2["oldKey"]
                            oldv2 = 0 if r2["isOld0"] else r2["oldValue"]
                        else:
                            st2 = dict(exit_leaves[key2])
                            enew = dict(st2)
                            enew["balance"] += eff_amount3
                            r2 = exit_tree.update(key2, db.hash_state(enew))
                        exit_leaves[key2] = enew
                        sib2 = r2["siblings"]
                    else:
                        rcur = receiver_leaf(new1)
                        if not on and to_idx == 0:
                            # transferToEthAddr / transferToBjj (src/rollup-tx.circom:253-276): the signed receiver must match
                            # 0xFF..FF selects transferToBjj, and still has to equal the leaf's ethAddr (Bjj-only accounts hold 0xFF..FF)
                            te = tx.get("toEthAddr", 0)
                            if te != rcur["ethAddr"]:
                                raise ValueError("toEthAddr does not match the receiver leaf (the circuit rejects this tx)")
                            if te == ETH_ADDR_ANY and (tx.get("toBjjAy", 0) != rcur["ay"] or tx.get("toBjjSign", 0) != rcur["sign"]):
                                raise ValueError("toBjj does not match the receiver leaf (the circuit rejects this tx)")
                        st2 = dict(rcur)
                        rnew = dict(rcur)
                        rnew["balance"] += eff_amount3
                        r2 = db.state.update(key2, db.hash_state(rnew))
                        db.leaves[key2] = rnew
                        sib2 = r2["siblings"]
                elif not on:
                    st2["tokenID"] = token   # processor 2 is a NOP, but the L2 receiver-token check still compares tokenID2 (src/rollup-tx.circom:270-274)
                if not on:
                    if "signer" in tx:
                        sig = tx["signer"].sign_msg(build_hash_sig(tx, db.chain_id))
                    else:
                        sig = {k: tx.get(k, 0) for k in ("r8x", "r8y", "s")}
            txc = build_tx_compressed_data(tx, db.chain_id) if not on else (
                CONST_SIG | (db.chain_id << 32) | (from_idx << 48) | (to_idx << 96) | (token << 144))
            inp["txCompressedData"].append(txc)
            inp["amountF"].append(amount_f)
            inp["txCompressedDataV2"].append(0 if on else build_tx_compressed_data_v2(tx))
            inp["fromIdx"].append(from_idx); inp["auxFromIdx"].append(aux_from)
            inp["toIdx"].append(to_idx); inp["auxToIdx"].append(aux_to)
            inp["toBjjAy"].append(tx.get("toBjjAy", 0)); inp["toEthAddr"].append(tx.get("toEthAddr", 0))
            inp["maxNumBatch"].append(tx.get("maxNumBatch", 0)); inp["onChain"].append(on); inp["newAccount"].append(new_account)
            inp["rqOffset"].append(tx.get("rqOffset", 0)); inp["rqTxCompressedDataV2"].append(tx.get("rqTxCompressedDataV2", 0))
            inp["rqToEthAddr"].append(tx.get("rqToEthAddr", 0)); inp["rqToBjjAy"].append(tx.get("rqToBjjAy", 0))
            inp["s"].append(sig["s"]); inp["r8x"].append(sig["r8x"]); inp["r8y"].append(sig["r8y"])
            inp["loadAmountF"].append(load_f); inp["fromEthAddr"].append(from_eth)
            inp["fromBjjCompressed"].append([(bjj >> k) & 1 for k in range(256)])
            for nm, st in (("1", st1), ("2", st2)):
                for f in ("tokenID", "nonce", "sign", "balance", "ay", "ethAddr"):
                    inp[f + nm].append(st[f])
            inp["siblings1"].append(self._pad(sib1)); inp["siblings2"].append(self._pad(sib2))
            inp["isOld0_1"].append(isold1); inp["oldKey1"].append(oldk1); inp["oldValue1"].append(oldv1)
            inp["isOld0_2"].append(isold2); inp["oldKey2"].append(oldk2); inp["oldValue2"].append(oldv2)
            inp["newExit"].append(new_exit)
            self.tx_meta.append({"isAmountNullified": is_nullified, "sigL2Hash": 0 if on else build_hash_sig(tx, db.chain_id),
                                 "stateRoot": db.state.root, "exitRoot": exit_tree.root, "accFee": list(acc_fee)})
            if i < nTx - 1:
                inp["imOnChain"].append(on); inp["imOutIdx"].append(db.last_idx)
                inp["imStateRoot"].append(db.state.root); inp["imExitRoot"].append(exit_tree.root)
                inp["imAccFeeOut"].append(list(acc_fee))
        # fee transactions (src/fee-tx.circom, src/rollup-main.circom:393-431)
        inp["imInitStateRootFee"] = db.state.root
        inp["imFinalAccFee"] = list(acc_fee)
        idxs = list(self.fee_idxs) + [0] * (F - len(self.fee_idxs))
        inp["feeIdxs"] = idxs
        for k in ("tokenID3", "nonce3", "sign3", "balance3", "ay3", "ethAddr3", "siblings3", "imStateRootFee"):
            inp[k] = []
        for j in range(F):
            st3, sib3 = {"tokenID": 0, "nonce": 0, "sign": 0, "balance": 0, "ay": 0, "ethAddr": 0}, []
            if idxs[j]:
                cur = db.leaves[idxs[j]]
                if cur["tokenID"] != plan[j]:
                    raise ValueError("fee idx token mismatch")
                st3 = dict(cur)
                new = dict(cur)
                new["balance"] += acc_fee[j]
                r3 = db.state.update(idxs[j], db.hash_state(new))
                db.leaves[idxs[j]] = new
                sib3 = r3["siblings"]
            for f, nm in (("tokenID", "tokenID3"), ("nonce", "nonce3"), ("sign", "sign3"), ("balance", "balance3"), ("ay", "ay3"), ("ethAddr", "ethAddr3")):
                inp[nm].append(st3[f])
            inp["siblings3"].append(self._pad(sib3))
            if j < F - 1:
                inp["imStateRootFee"].append(db.state.root)
        if db.lazy:
            # every hash of the batch in nLevels + 3 device launches; then references -> values
            val = db.flush(extra_trees=(exit_tree,))
            for k in ("siblings1", "siblings2", "siblings3"):
                inp[k] = [[val(x) for x in row] for row in inp[k]]
            for k in ("oldValue1", "oldValue2", "imStateRoot", "imExitRoot", "imStateRootFee"):
                inp[k] = [val(x) for x in inp[k]]
            inp["imInitStateRootFee"] = val(inp["imInitStateRootFee"])
            for m in self.tx_meta:
                m["stateRoot"], m["exitRoot"] = val(m["stateRoot"]), val(m["exitRoot"])
        self.input = inp
        self.new_state_root = db.state.root
        self.new_last_idx = db.last_idx
        self.new_exit_root = exit_tree.root
        self.exit_tree, self.exit_leaves = exit_tree, exit_leaves
        db.exit_trees[self.current_num_batch] = (exit_tree, exit_leaves)
        db.num_batch = self.current_num_batch
        self.built = True
        return self

    def get_input(self):
        return self.input

    # data-availability strings and the global hash (src/hash-inputs.circom:117-184)
    def get_hash_inputs(self):
        L, F, nTx = self.L, self.F, self.nTx
        inp = self.input
        bits = []

        def be(v, n):
            bits.extend((v >> (n - 1 - k)) & 1 for k in range(n))
        be(inp["oldLastIdx"], 48); be(self.new_last_idx, 48); be(inp["oldStateRoot"], 256); be(self.new_state_root, 256); be(self.new_exit_root, 256)
        for i in range(self.maxL1):
            on = inp["onChain"][i] if i < nTx else 0
            if on:
                bjj = sum(b << k for k, b in enumerate(inp["fromBjjCompressed"][i]))
                txc = inp["txCompressedData"][i]
                be(inp["fromEthAddr"][i], 160); be(bjj, 256); be((txc >> 48) & ((1 << 48) - 1), 48); be(inp["loadAmountF"][i], 40)
                be(inp["amountF"][i], 40); be((txc >> 144) & 0xFFFFFFFF, 32); be((txc >> 96) & ((1 << 48) - 1), 48)
            else:
                bits.extend([0] * 624)
        for i in range(nTx):
            txc = inp["txCompressedData"][i]
            on = inp["onChain"][i]
            frm = (txc >> 48) & ((1 << 48) - 1)
            to = (txc >> 96) & ((1 << 48) - 1)
            final_to = inp["auxToIdx"][i] if (not on and to == 0) else to
            be(frm, L); be(final_to, L)
            be(0 if self.tx_meta[i]["isAmountNullified"] else inp["amountF"][i], 40)
            be(0 if on else (txc >> 216) & 0xFF, 8)
        for j in range(F):
            be(inp["feeIdxs"][j], L)
        be(inp["globalChainID"], 16); be(inp["currentNumBatch"], 32)
        return int.from_bytes(sha256_bits(bits), "big") % P

    # reference test/helpers/helpers.js:45-137 getSingleTxInput
    def get_single_tx_input(self, i):
        inp, F = self.input, self.F
        txc = inp["txCompressedData"][i]
        r = {
            "feePlanTokens": inp["feePlanTokens"], "accFeeIn": [0] * F,
            "futureTxCompressedDataV2": [inp["txCompressedDataV2"][i + j + 1] if i + j + 1 < self.nTx else 0 for j in range(3)],
            "pastTxCompressedDataV2": [inp["txCompressedDataV2"][i - j - 1] if i - j - 1 >= 0 else 0 for j in range(4)],
            "futureToEthAddr": [inp["toEthAddr"][i + j + 1] if i + j + 1 < self.nTx else 0 for j in range(3)],
            "pastToEthAddr": [inp["toEthAddr"][i - j - 1] if i - j - 1 >= 0 else 0 for j in range(4)],
            "futureToBjjAy": [inp["toBjjAy"][i + j + 1] if i + j + 1 < self.nTx else 0 for j in range(3)],
            "pastToBjjAy": [inp["toBjjAy"][i - j - 1] if i - j - 1 >= 0 else 0 for j in range(4)],
            "fromIdx": (txc >> 48) & ((1 << 48) - 1), "auxFromIdx": inp["auxFromIdx"][i], "toIdx": (txc >> 96) & ((1 << 48) - 1),
            "auxToIdx": inp["auxToIdx"][i], "toBjjAy": inp["toBjjAy"][i], "toBjjSign": (txc >> 224) & 1, "toEthAddr": inp["toEthAddr"][i],
            "amount": float2fix(inp["amountF"][i]), "tokenID": (txc >> 144) & 0xFFFFFFFF, "nonce": (txc >> 176) & ((1 << 40) - 1),
            "userFee": (txc >> 216) & 0xFF, "rqOffset": inp["rqOffset"][i], "onChain": inp["onChain"][i], "newAccount": inp["newAccount"][i],
            "rqTxCompressedDataV2": inp["rqTxCompressedDataV2"][i], "rqToEthAddr": inp["rqToEthAddr"][i], "rqToBjjAy": inp["rqToBjjAy"][i],
            "sigL2Hash": self.tx_meta[i]["sigL2Hash"],
            "s": inp["s"][i], "r8x": inp["r8x"][i], "r8y": inp["r8y"][i], "fromEthAddr": inp["fromEthAddr"][i],
            "fromBjjCompressed": inp["fromBjjCompressed"][i], "loadAmountF": inp["loadAmountF"][i],
            "oldStateRoot": inp["imStateRoot"][i - 1] if i > 0 else inp["oldStateRoot"],
            "oldExitRoot": inp["imExitRoot"][i - 1] if i > 0 else 0,
        }
        for k in ("tokenID1", "nonce1", "sign1", "balance1", "ay1", "ethAddr1", "siblings1", "isOld0_1", "oldKey1", "oldValue1", "tokenID2",
                  "nonce2", "sign2", "balance2", "newExit", "ay2", "ethAddr2", "siblings2", "isOld0_2", "oldKey2", "oldValue2"):
            r[k] = inp[k][i]
        prev = self.tx_meta[i - 1]["accFee"] if i > 0 else [0] * F
        out = {"accFeeOut": [a - b for a, b in zip(self.tx_meta[i]["accFee"], prev)],
               "newStateRoot": self.tx_meta[i]["stateRoot"], "newExitRoot": self.tx_meta[i]["exitRoot"],
               "isAmountNullified": self.tx_meta[i]["isAmountNullified"]}
        return r, out


def withdraw_input(batch, idx, n_levels):
    """Inputs of Withdraw(nLevels) for an exit leaf of a built batch (reference test/withdraw.test.js:39-157)."""
    tree, leaves = batch.exit_tree, batch.exit_leaves
    st = leaves[idx]
    f = tree.find(idx)
    assert f["found"]
    sib = list(f["siblings"]) + [0] * (n_levels + 1 - len(f["siblings"]))
    inp = {"rootExit": tree.root, "ethAddr": st["ethAddr"], "tokenID": st["tokenID"], "balance": st["balance"], "idx": idx, "sign": st["sign"],
           "ay": st["ay"], "siblingsState": sib}
    bits = []

    def be(v, n):
        bits.extend((v >> (n - 1 - k)) & 1 for k in range(n))
    be(tree.root, 256); be(st["ethAddr"], 160); be(st["tokenID"], 32); be(st["balance"], 192); be(idx, 48)
    by = bytes(sum(bits[8 * i + k] << (7 - k) for k in range(8)) for i in range(len(bits) // 8))
    return inp, int.from_bytes(hashlib.sha256(by).digest(), "big") % P


class ExitTreeFixture:
    """An exit tree built directly (what a batch of `n_leaves` exits leaves behind; reference test/withdraw.test.js:39-157 reaches it
    through four exit transactions): exit_tree / exit_leaves as withdraw_input() reads them. device=N hashes the tree on GPU N."""

    def __init__(self, n_leaves, seed=0x57495448, first_idx=256, n_keys=8, device=None, dag_evaluator=None):
        import random
        rng = random.Random(seed)
        lazy = device is not None or dag_evaluator is not None
        hasher = DagHasher(dag_evaluator or _device_dag_evaluator(device)) if lazy else host()
        keys = [Account(seed * 1000 + i) for i in range(n_keys)]
        self.exit_tree, self.exit_leaves = SMT(hasher), {}
        for i in range(n_leaves):
            a = keys[rng.randrange(n_keys)]
            st = {"tokenID": rng.randrange(1, 1 << 32), "nonce": 0, "sign": a.sign, "balance": rng.randrange(1, 1 << 192), "ay": a.ay, "ethAddr": a.eth_addr}
            self.exit_tree.insert(first_idx + i, hash_state(st, hasher))
            self.exit_leaves[first_idx + i] = st
        if lazy:
            self.exit_tree.rekey(hasher.resolve())


def synthetic_batch(n_tx, n_levels, max_l1, max_fee, seed=0x48455A31, n_accounts=None, n_keys=8, exits=0, device=None, dag_evaluator=None, first_idx=256,
                    base=None, dense=False):
    """Seeded synthetic batch following reference tools/generate-input.js:61-109 and
    tools/helpers/gen-inputs-utils.js: pre-populated accounts (token 1), then one batch of `max_l1` L1
    createAccountDeposit txs followed by signed L2 transfers of 20 % of the sender balance with
    userFee 176 (plus `exits` L2 exits), one fee token and one fee receiver."""
    import random
    rng = random.Random(seed)
    if base is None and dense and n_accounts and n_accounts >= 16 and n_accounts & (n_accounts - 1) == 0:
        # the same pre-populated state built in bulk (DenseState: 3 hashes per account, one library call per tree level) instead of
        # account by account (a path of ~log2(n) hashes per insertion): same tree shape, 5x fewer hashes -- what bench.py's
        # batch builders use (8192 accounts per batch)
        base = DenseState.build(n_accounts.bit_length() - 1, seed=seed, first_idx=first_idx, n_keys=n_keys)
    if base is not None:
        return _synthetic_batch_on_base(rng, base, n_tx, n_levels, max_l1, max_fee, seed, n_keys, exits, device, dag_evaluator)
    db = RollupDB(chain_id=1, device=device, dag_evaluator=dag_evaluator, first_idx=first_idx)
    keys = [Account(seed * 1000 + i) for i in range(n_keys)]
    n_accounts = n_accounts if n_accounts is not None else max(2, min(4 * n_tx, 4096))
    owner = {}
    # pre-population (direct state construction, equivalent to earlier deposit batches)
    for _ in range(n_accounts):
        db.last_idx += 1
        a = keys[rng.randrange(n_keys)]
        bal = float2fix(floor_fix2float(rng.randrange(1 << 96)))
        st = {"tokenID": 1, "nonce": 0, "sign": a.sign, "balance": bal, "ay": a.ay, "ethAddr": a.eth_addr}
        db.state.insert(db.last_idx, db.hash_state(st))
        db.leaves[db.last_idx] = st
        owner[db.last_idx] = a
    bb = db.build_batch(n_tx, n_levels, max_l1, max_fee)
    n_l1 = min(max_l1, n_tx)
    for _ in range(n_l1):
        a = keys[rng.randrange(n_keys)]
        bb.add_tx({"fromIdx": 0, "loadAmountF": floor_fix2float(rng.randrange(1 << 96)), "tokenID": 1, "fromBjjCompressed": a.bjj_compressed,
                   "fromEthAddr": a.eth_addr, "toIdx": 0, "onChain": 1})
    tmp = {}
    idxs = sorted(owner)
    for t in range(n_tx - n_l1):
        frm = idxs[rng.randrange(len(idxs))]
        to = idxs[rng.randrange(len(idxs))]
        bal, nonce = tmp.get(frm, (db.leaves[frm]["balance"], db.leaves[frm]["nonce"]))
        amount = float2fix(floor_fix2float(bal * 20 // 100))
        is_exit = t < exits
        tx = {"fromIdx": frm, "toIdx": EXIT_IDX if is_exit else to, "amount": amount, "tokenID": 1, "userFee": 176, "nonce": nonce, "onChain": 0,
              "signer": owner[frm]}
        bb.add_tx(tx)
        nb = bal - amount - compute_fee(amount, 176)
        tmp[frm] = (nb, nonce + 1)
        if not is_exit and to != frm:
            tb, tn = tmp.get(to, (db.leaves[to]["balance"], db.leaves[to]["nonce"]))
            tmp[to] = (tb + amount, tn)
        elif not is_exit and to == frm:
            tmp[frm] = (nb + amount, nonce + 1)
    bb.add_token(1)
    bb.add_fee_idx(idxs[rng.randrange(len(idxs))])
    bb.build()
    return bb


def _synthetic_batch_on_base(rng, base, n_tx, n_levels, max_l1, max_fee, seed, n_keys, exits, device, dag_evaluator):
    """synthetic_batch's recipe on a shared pre-populated DenseState: the batch's own L1 keys and transactions come from `seed`,
    senders and receivers are drawn from the base's N accounts (their keys are the base's)."""
    db = RollupDB(chain_id=1, device=device, dag_evaluator=dag_evaluator, base=base)
    keys = [Account(seed * 1000 + i) for i in range(n_keys)]
    bkeys = base.keys()
    bb = db.build_batch(n_tx, n_levels, max_l1, max_fee)
    n_l1 = min(max_l1, n_tx)
    for _ in range(n_l1):
        a = keys[rng.randrange(n_keys)]
        bb.add_tx({"fromIdx": 0, "loadAmountF": floor_fix2float(rng.randrange(1 << 96)), "tokenID": 1, "fromBjjCompressed": a.bjj_compressed,
                   "fromEthAddr": a.eth_addr, "toIdx": 0, "onChain": 1})
    tmp = {}
    pick = lambda: base.first_idx + rng.randrange(base.N)   # noqa: E731
    for t in range(n_tx - n_l1):
        frm, to = pick(), pick()
        bal, nonce = tmp.get(frm, (db.leaves[frm]["balance"], db.leaves[frm]["nonce"]))
        amount = float2fix(floor_fix2float(bal * 20 // 100))
        is_exit = t < exits
        bb.add_tx({"fromIdx": frm, "toIdx": EXIT_IDX if is_exit else to, "amount": amount, "tokenID": 1, "userFee": 176, "nonce": nonce, "onChain": 0,
                   "signer": bkeys[int(base.key_idx[frm - base.first_idx])]})
        nb = bal - amount - compute_fee(amount, 176)
        tmp[frm] = (nb, nonce + 1)
        if not is_exit and to != frm:
            tb, tn = tmp.get(to, (db.leaves[to]["balance"], db.leaves[to]["nonce"]))
            tmp[to] = (tb + amount, tn)
        elif not is_exit and to == frm:
            tmp[frm] = (nb + amount, nonce + 1)
    bb.add_token(1)
    bb.add_fee_idx(pick())
    bb.build()
    return bb
