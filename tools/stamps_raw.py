"""Dev tool: the tile engine's phase stamps as a chip-wide timeline (who is late, and by how much).
usage: python tools/stamps_raw.py [n] [steps]        (runs one asynchronous call under GRAPHLEARN_AMD_RES_STAMPS and reads the raw stamps)
       python tools/stamps_raw.py --read FILE        (only analyse a file written through GRAPHLEARN_AMD_RES_STAMPS_RAW)
Per stamped iteration (8 .. 23 of the launch), relative to the earliest pass start of that iteration: when the waves start and end
their pass, store their records, when the owners have seen all records and stored the node values, and when the consumers have
seen them (= the start of the next pass).  Then who is slow: per wave slot, per XCD (workgroup id mod 8), the slowest waves."""
import os, sys
import numpy as np


def read(path):
    raw = np.fromfile(path, dtype=np.uint64)
    out, pos = [], 0
    while pos < raw.size:
        assert raw[pos] == 0x544C5354414D5053, "bad header"
        NWG, nb, TW = int(raw[pos + 1]), int(raw[pos + 2]), int(raw[pos + 3])
        pos += 4
        szA, szB, szC, szD = NWG * 16 * 6, NWG * 16 * TW * 2, NWG * TW * 8, NWG * 16 * TW * 4
        A = raw[pos:pos + szA].reshape(NWG, 16, 6).astype(np.int64); pos += szA
        B = raw[pos:pos + szB].reshape(NWG, 16, TW, 2).astype(np.int64); pos += szB
        C = raw[pos:pos + szC].reshape(NWG, TW, 8).astype(np.int64); pos += szC
        D = raw[pos:pos + szD].reshape(NWG, 16, TW, 4).astype(np.int64); pos += szD
        out.append((NWG, nb, TW, A, B, C, D))
    return out


def pct(x, qs=(50, 90, 99, 100)):
    x = np.asarray(x, dtype=np.float64)
    x = x[np.isfinite(x)]
    if x.size == 0:
        return "   -  " * len(qs)
    return " ".join(f"{np.percentile(x, q) / 100:6.2f}" for q in qs)


def analyse(NWG, nb, TW, A, B, C, D, label=""):
    print(f"== launch {label}: {NWG} workgroups, {nb} node blocks")
    ok = B[:, :, :, 1] > 0                                   # waves with a pass in that slot
    dur = np.where(ok, B[:, :, :, 1] - B[:, :, :, 0], 0)
    ks = [k for k in range(1, 15) if ok[:, k].any() and ok[:, k + 1].any()]
    if not ks:
        print("   no stamped iterations")
        return
    print("   timeline per iteration, us after the earliest pass start of the chip: p50 p90 p99 max")
    rows = {name: [] for name in ("pass start", "pass end", "records stored", "owner: records seen", "owner: node values stored",
                                 "node values seen (next)", "next pass start")}
    spans = []
    for k in ks:
        t0 = B[:, k, :, 0][ok[:, k]].min()
        rows["pass start"].append(B[:, k, :, 0][ok[:, k]] - t0)
        rows["pass end"].append(B[:, k, :, 1][ok[:, k]] - t0)
        d1 = D[:, k, :, 1]; rows["records stored"].append(d1[d1 > 0] - t0)
        d2 = D[:nb, k, :, 2]; rows["owner: records seen"].append(d2[d2 > 0] - t0)
        d3 = D[:nb, k, 0, 3]; rows["owner: node values stored"].append(d3[d3 > 0] - t0)
        d0 = D[:, k + 1, :, 0]; rows["node values seen (next)"].append(d0[d0 > 0] - t0)
        rows["next pass start"].append(B[:, k + 1, :, 0][ok[:, k + 1]] - t0)
        spans.append(B[:, k + 1, :, 0][ok[:, k + 1]].min() - t0)
    for name, v in rows.items():
        print(f"   {name:28s} {pct(np.concatenate(v))}")
    print(f"   iteration (earliest start to earliest start): {np.mean(spans) / 100:.2f} us")
    # who is slow
    d = np.where(ok[:, ks], dur[:, ks], np.nan)
    print("   pass duration by wave slot:", " ".join(f"{np.nanmean(d[:, :, v]) / 100:.2f}" for v in range(TW)))
    print("   pass duration by XCD (workgroup mod 8):", " ".join(f"{np.nanmean(d[x::8]) / 100:.2f}" for x in range(8)))
    per_wave = np.nanmean(d, axis=1)                         # (NWG, TW)
    flat = [(per_wave[g, v], g, v) for g in range(NWG) for v in range(TW) if np.isfinite(per_wave[g, v])]
    flat.sort(reverse=True)
    print("   slowest waves (mean us, workgroup, wave):", ", ".join(f"{a / 100:.2f} wg{g}.{v}" for a, g, v in flat[:12]))
    print("   fastest waves:", ", ".join(f"{a / 100:.2f} wg{g}.{v}" for a, g, v in flat[-6:]))
    it_max = np.nanmax(np.nanmax(d, axis=2), axis=0)
    print("   slowest pass of the chip per iteration:", " ".join(f"{a / 100:.2f}" for a in it_max))
    # how late the latest pass end of a workgroup is against its first (the publish waits for it)
    kn = [k + 1 for k in ks]
    okk, okn = ok[:, ks], ok[:, kn]
    Bk, Bn, Dk, Dn = B[:, ks], B[:, kn], D[:, ks], D[:, kn]
    pe = np.where(okk, Bk[..., 1], 0)
    spread = pe.max(axis=2) - np.where(okk, Bk[..., 1], np.iinfo(np.int64).max).min(axis=2)
    print(f"   inside a workgroup, last pass end - first pass end: mean {np.mean(spread) / 100:.2f} max {np.max(spread) / 100:.2f}")
    # records stored relative to the workgroup's last pass end
    d1 = Dk[..., 1]
    lag = np.where(d1 > 0, d1 - pe.max(axis=2, keepdims=True), np.nan)
    print(f"   records stored after the workgroup's last pass end: mean {np.nanmean(lag) / 100:.2f} max {np.nanmax(lag) / 100:.2f}")
    # owners: from the last record seen to the node values stored
    d2, d3 = Dk[:nb, :, :, 2], Dk[:nb, :, 0, 3]
    seen = np.where(d2 > 0, d2, 0).max(axis=2)
    upd = np.where((d3 > 0) & (seen > 0), d3 - seen, np.nan)
    print(f"   owners, all records seen -> node values stored: mean {np.nanmean(upd) / 100:.2f} max {np.nanmax(upd) / 100:.2f}")
    # owners: the workgroup's own records stored -> all records of its block seen (the wait for the slowest producer + a hop)
    own_pub = np.where(Dk[:nb, :, :, 1] > 0, Dk[:nb, :, :, 1], 0).max(axis=2)
    wait = np.where((seen > 0) & (own_pub > 0), seen - own_pub, np.nan)
    print(f"   owners, own records stored -> all records seen: mean {np.nanmean(wait) / 100:.2f} max {np.nanmax(wait) / 100:.2f}")
    # consumers: node values seen -> pass start
    d0, st = Dn[..., 0], Bn[..., 0]
    seen_wg = np.where(d0 > 0, d0, 0).max(axis=2, keepdims=True)
    gap = np.where((seen_wg > 0) & okn, st - seen_wg, np.nan)
    print(f"   consumers, last node value seen in the workgroup -> pass start: mean {np.nanmean(gap) / 100:.2f} max {np.nanmax(gap) / 100:.2f}")
    # the two hops: the latest record stored for a block -> its owner has seen all; the owner's store -> the latest consumer has seen it
    hop2 = []
    for i, k in enumerate(ks):
        lastseen = np.where(d0[:, i] > 0, d0[:, i], 0).max()
        laststore = np.where(d3[:, i] > 0, d3[:, i], 0).max()
        if lastseen and laststore:
            hop2.append(lastseen - laststore)
    if hop2:
        print(f"   latest owner's node values stored -> latest consumer has seen its values: mean {np.mean(hop2) / 100:.2f}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--read":
        for i, rec in enumerate(read(sys.argv[2])):
            analyse(*rec, label=str(i))
        sys.exit(0)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    path = os.environ.setdefault("GRAPHLEARN_AMD_RES_STAMPS_RAW", "/tmp/tile_stamps.raw")
    os.environ["GRAPHLEARN_AMD_RES_STAMPS"] = "1"
    if os.path.exists(path):
        os.remove(path)
    from graph_learning_amd import _lib
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    ctx = _lib.default_context()
    z = np.random.default_rng(n).uniform(0.05, 3.0, size=n * (n - 1) // 2)
    with _lib.LogDegreeSession(n, ctx) as s:
        s.set_distances(z)
        s.init(1, 1, 0.5, None)
        s.iterate_async(steps)
        s.dual()
        s.iterate_async(steps)
        s.dual()
        s.iterate(steps, 0.0)
    recs = read(path)
    for i, rec in enumerate(recs):
        analyse(*rec, label=["async 0", "async 1", "sync (norms)"][i] if i < 3 else str(i))
