import numpy as np
M32 = np.uint64(0xffffffff)
def mix32(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & M32; x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & M32; x ^= x >> np.uint64(16)
    return x
def mul24(a, b):
    return ((a & np.uint64(0xffffff)) * (np.uint64(b) & np.uint64(0xffffff))) & M32
def hash24(x, C1=0x9E3779, C2=0x85EBCB, s1=13, s2=11):
    x = x.astype(np.uint64)
    h = mul24(x, C1); h ^= h >> np.uint64(s1)
    h = mul24(h, C2); h ^= h >> np.uint64(s2)
    return h
def planes(gen, nplanes=64, Q=256, K=256, seed=12345):
    # bytes[plane, q, k]: uniform 8-bit per element; quad = q*(K/4) + k/4, word per quad, byte k&3
    out = np.empty((nplanes, Q, K), np.uint8)
    quad = (np.arange(Q)[:, None] * (K // 4) + np.arange(K // 4)[None, :]).astype(np.uint64)
    for p in range(nplanes):
        key = mix32(np.array([seed ^ (p * 0x85EBCA6B & 0xffffffff)], dtype=np.uint64))[0]
        w = gen(quad ^ key)
        for b in range(4):
            out[p, :, b::4] = ((w >> np.uint64(8 * b)) & np.uint64(0xff)).astype(np.uint8)
    return out
def battery(name, u8, t=26):
    drop = (u8 < t).astype(np.float64)     # P = t/256
    p = t / 256
    n = drop.size
    z_mean = (drop.mean() - p) / np.sqrt(p * (1 - p) / n)
    # byte-value chi-square (256 bins)
    cnt = np.bincount(u8.ravel(), minlength=256); e = n / 256
    chi = ((cnt - e) ** 2 / e).sum(); z_chi = (chi - 255) / np.sqrt(2 * 255)
    d = drop - p; v = p * (1 - p)
    def corr(a, b):
        return (a * b).mean() / v * np.sqrt(a.size)      # z-score of the correlation
    lags_k = [corr(d[:, :, :-l], d[:, :, l:]) for l in (1, 2, 3, 4, 5, 8, 16, 64)]
    lags_q = [corr(d[:, :-l, :], d[:, l:, :]) for l in (1, 2, 4, 16)]
    lags_p = [corr(d[:-l], d[l:]) for l in (1, 2, 8)]
    # per-row and per-column keep rates: variance of row means vs binomial
    rm = drop.mean(axis=2); z_rows = (rm.var() / (v / drop.shape[2]) - 1) * np.sqrt(rm.size / 2)
    cm = drop.mean(axis=1); z_cols = (cm.var() / (v / drop.shape[1]) - 1) * np.sqrt(cm.size / 2)
    pm = drop.mean(axis=(1, 2)); z_pl = (pm.var() / (v / (drop.shape[1] * drop.shape[2])) - 1) * np.sqrt(pm.size / 2)
    print(f"{name:12s} mean z {z_mean:+.2f}  chi z {z_chi:+.2f}  key-lags {' '.join(f'{x:+.1f}' for x in lags_k)}  q-lags {' '.join(f'{x:+.1f}' for x in lags_q)}  plane-lags {' '.join(f'{x:+.1f}' for x in lags_p)}  rows {z_rows:+.1f} cols {z_cols:+.1f} planes {z_pl:+.1f}")
rng = np.random.default_rng(0)
battery("numpy", rng.integers(0, 256, (64, 256, 256), dtype=np.uint8))
battery("mix32", planes(mix32))
battery("hash24", planes(hash24))
for (c1, c2, s1, s2) in [(0x9E3779, 0x85EBCB, 12, 12), (0xB5297B, 0x68E31D, 13, 11), (0x9E3779, 0x85EBCB, 11, 13), (0xC2B2AF, 0x27D4EB, 13, 11)]:
    battery(f"h24 {s1},{s2}", planes(lambda x: hash24(x, c1, c2, s1, s2)))
battery("1mul", planes(lambda x: (lambda h: h ^ (h >> np.uint64(13)))(mul24(x, 0x9E3779))))

print("---- stronger battery (256 planes x 256 x 256 = 16.8 M decisions)")
def chi2d(a, b, bins=16):
    ia = (a >> 4).astype(np.int64); ib = (b >> 4).astype(np.int64)
    c = np.bincount((ia * bins + ib).ravel(), minlength=bins * bins); e = ia.size / (bins * bins)
    chi = ((c - e) ** 2 / e).sum(); k = bins * bins - 1
    return (chi - k) / np.sqrt(2 * k)
def battery2(name, u8):
    zs = {}
    for t in (26, 25, 13, 77):
        p = t / 256; drop = (u8 < t); n = drop.size
        zs[f"mean{t}"] = (drop.mean() - p) / np.sqrt(p * (1 - p) / n)
    zs["b0b1"] = chi2d(u8[:, :, 0::4], u8[:, :, 1::4]); zs["b1b2"] = chi2d(u8[:, :, 1::4], u8[:, :, 2::4]); zs["b2b3"] = chi2d(u8[:, :, 2::4], u8[:, :, 3::4]); zs["b0b3"] = chi2d(u8[:, :, 0::4], u8[:, :, 3::4])
    zs["w,w+1"] = chi2d(u8[:, :, 0:-4:4], u8[:, :, 4::4]); zs["w3,w+1_0"] = chi2d(u8[:, :, 3:-4:4], u8[:, :, 4::4])
    zs["q,q+1"] = chi2d(u8[:, :-1, :], u8[:, 1:, :]); zs["p,p+1"] = chi2d(u8[:-1], u8[1:])
    lowbits = u8 & 15
    zs["low4 k,k+1"] = chi2d(lowbits[:, :, :-1] << 4, lowbits[:, :, 1:] << 4)
    d = (u8 < 26).astype(np.float64) - 26 / 256; v = (26 / 256) * (1 - 26 / 256)
    for l in (1, 2, 3, 4, 7, 8, 12, 16, 32, 64, 128):
        zs[f"k{l}"] = (d[:, :, :-l] * d[:, :, l:]).mean() / v * np.sqrt(d[:, :, l:].size)
    for l in (1, 2, 3, 4, 8, 16, 32):
        zs[f"q{l}"] = (d[:, :-l, :] * d[:, l:, :]).mean() / v * np.sqrt(d[:, l:, :].size)
    for l in (1, 2, 4, 8, 64):
        zs[f"p{l}"] = (d[:-l] * d[l:]).mean() / v * np.sqrt(d[l:].size)
    # diagonal (q+1, k+1) and anti-diagonal
    zs["diag"] = (d[:, :-1, :-1] * d[:, 1:, 1:]).mean() / v * np.sqrt(d[:, 1:, 1:].size)
    zs["adiag"] = (d[:, :-1, 1:] * d[:, 1:, :-1]).mean() / v * np.sqrt(d[:, 1:, 1:].size)
    worst = max(zs, key=lambda k: abs(zs[k]))
    nbad = sum(abs(z) > 3 for z in zs.values())
    print(f"{name:12s} {len(zs)} statistics: worst {worst} z = {zs[worst]:+.2f}; |z| > 3: {nbad}; rms z {np.sqrt(np.mean(np.square(list(zs.values())))):.2f}")
    return zs
battery2("numpy", rng.integers(0, 256, (256, 256, 256), dtype=np.uint8))
battery2("mix32", planes(mix32, 256))
for (c1, c2, s1, s2) in [(0x9E3779, 0x85EBCB, 12, 12), (0x9E3779, 0x85EBCB, 13, 11), (0xB5297B, 0x68E31D, 12, 12), (0xC2B2AF, 0x27D4EB, 12, 12)]:
    battery2(f"h24 {c1:x} {s1},{s2}", planes(lambda x: hash24(x, c1, c2, s1, s2), 256))

print("---- keyed form: h = mul24(quad ^ k1, C1); h = h ^ (h >> 12) ^ k2; h = mul24(h, C2); h ^= h >> 12")
def hash24k(quad, k1, k2, C1=0x9E3779, C2=0x85EBCB):
    h = mul24(quad.astype(np.uint64) ^ np.uint64(k1), C1); h = h ^ (h >> np.uint64(12)) ^ np.uint64(k2)
    h = mul24(h, C2); h ^= h >> np.uint64(12)
    return h
def planes_k(nplanes=256, Q=256, K=256, seed=777, same_k1=False):
    out = np.empty((nplanes, Q, K), np.uint8)
    quad = (np.arange(Q)[:, None] * (K // 4) + np.arange(K // 4)[None, :]).astype(np.uint64)
    for p in range(nplanes):
        k1 = int(mix32(np.array([seed ^ (0 if same_k1 else p * 0x85EBCA6B & 0xffffffff)], dtype=np.uint64))[0])
        k2 = int(mix32(np.array([(k1 ^ 0xB5297A4D) + p * 0x9E3779B9 & 0xffffffff], dtype=np.uint64))[0])
        w = hash24k(quad, k1, k2)
        for b in range(4):
            out[p, :, b::4] = ((w >> np.uint64(8 * b)) & np.uint64(0xff)).astype(np.uint8)
    return out
battery2("keyed", planes_k())
battery2("keyed same k1", planes_k(same_k1=True))
battery2("keyed seed2", planes_k(seed=99991))
