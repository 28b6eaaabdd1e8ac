"""Offline experiment (CPU, numpy): how much row-slot efficiency a pairwise local search over adjacent 16-row groups
adds on top of the plan's Gray-rank order (0.872 -> 0.882 after one pass, 0.887 after six)."""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "markerless-robot-camera-calibration_amd", "synth.py"))
synth = importlib.util.module_from_spec(spec)
spec.loader.exec_module(synth)
def morton3(x, y, z, bits=18):
    out = np.zeros_like(x, dtype=np.uint64)
    for b in range(bits):
        out |= ((x >> b) & 1).astype(np.uint64) << np.uint64(3 * b)
        out |= ((y >> b) & 1).astype(np.uint64) << np.uint64(3 * b + 1)
        out |= ((z >> b) & 1).astype(np.uint64) << np.uint64(3 * b + 2)
    return out
def build(c, ts):
    V = len(c); bias = 1 << 17
    key = morton3(*(c.T.astype(np.int64) + bias))
    order = np.argsort(key); c = c[order]; key = key[order]
    act = np.zeros((27, V), bool); ki = 0
    for dz in (-1,0,1):
        for dy in (-1,0,1):
            for dx in (-1,0,1):
                q = c + np.array([dx,dy,dz]) * ts
                kq = morton3(*(q.T.astype(np.int64) + bias))
                pos = np.searchsorted(key, kq); pos[pos >= V] = V-1
                act[ki] = key[pos] == kq; ki += 1
    return c, act
def cls(k):
    dx, dy, dz = k % 3 - 1, (k // 3) % 3 - 1, k // 9 - 1
    return abs(dx) + abs(dy) + abs(dz)  # 3 corner,2 edge,1 face,0 centre
def eff(act, perm, g=16):
    V = act.shape[1]; a = act[:, perm]; pad = (-V) % g
    if pad: a = np.concatenate([a, np.zeros((27, pad), bool)], axis=1)
    t = a.reshape(27, -1, g).any(axis=2)
    return act.sum() / (t.sum() * g)
def key_from_order(act, bitorder):  # bitorder[0] = most significant
    m = np.zeros(act.shape[1], dtype=np.int64)
    for rank, k in enumerate(bitorder):
        m |= act[k].astype(np.int64) << (26 - rank)
    return m
def gray_decode(m):
    r = m.copy(); s = 1
    while s < 32:
        r ^= r >> s; s *= 2
    return r
n, L, scale = 200000, 2.4, 50
pts, rgb, lab = synth.gen_room(n, L, 0)
c = np.unique(np.floor(pts * scale).astype(np.int64), axis=0); ts = 1
c, act = build(c, ts); V = act.shape[1]
cur = sorted(range(27), key=lambda k: (-cls(k), k))
m = gray_decode(key_from_order(act, cur))
perm = np.argsort(m, kind="stable")
raw = np.zeros(V, dtype=np.int64)
for k in range(27): raw |= act[k].astype(np.int64) << k
pc_tab = np.array([bin(i).count("1") for i in range(1 << 16)], dtype=np.int64)
def popc(x): return pc_tab[x & 0xffff] + pc_tab[(x >> 16) & 0xffff]
masks = raw[perm]
pad = (-V) % 16
masks = np.concatenate([masks, np.zeros(pad, dtype=np.int64)])
G = masks.reshape(-1, 16).copy()
def cost(G): return popc(np.bitwise_or.reduce(G, axis=1)).sum()
P = popc(raw).sum()
print("start eff", P / (cost(G) * 16))
# local search: for adjacent group pairs, re-split 32 rows: seeds = the two rows with the largest hamming distance
rng = np.random.default_rng(0)
for it in range(6):
    for off in (0, 1):
        for g in range(off, len(G) - 1, 2):
            rows = np.concatenate([G[g], G[g + 1]])
            base = popc(np.bitwise_or.reduce(G[g])) + popc(np.bitwise_or.reduce(G[g + 1]))
            best = None
            for trial in range(4):
                if trial == 0:
                    # farthest pair seeds
                    x = rows[:, None] ^ rows[None, :]
                    d = popc(x); i, j = np.unravel_index(np.argmax(d), d.shape)
                else:
                    i, j = rng.choice(32, 2, replace=False)
                u1, u2 = rows[i], rows[j]; a, b = [i], [j]
                rest = [r for r in range(32) if r not in (i, j)]
                # assign in order of decreasing popcount
                rest.sort(key=lambda r: -popc(rows[r]))
                for r in rest:
                    c1 = popc(rows[r] & ~u1); c2 = popc(rows[r] & ~u2)
                    if len(a) >= 16: ch = 2
                    elif len(b) >= 16: ch = 1
                    else: ch = 1 if (c1 < c2 or (c1 == c2 and len(a) <= len(b))) else 2
                    if ch == 1: a.append(r); u1 |= rows[r]
                    else: b.append(r); u2 |= rows[r]
                cst = popc(u1) + popc(u2)
                if cst < base and (best is None or cst < best[0]): best = (cst, a, b)
            if best is not None:
                G[g] = rows[best[1]]; G[g + 1] = rows[best[2]]
    print("iter", it, "eff", P / (cost(G) * 16))
