"""profiles/rNN_pmc_traffic.md + profiles/traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of
`python bench.py --steps 1 --warmup 1 --no-cpu-baseline` (2 steps).

    python tools/pmc_table.py <fetch.db> <write.db> <out.md> <traffic.json> [dominant-kernel-prefix]

HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: on gfx950 FETCH_SIZE counts half of the bytes read (MI355X guide);
re-checked in every run on `chscale_bwd_kernel`, a pure streaming pass whose traffic is known: it reads two tensors of the encoder
output's size [B, 128V, L/4] (+ a [B, V, 128] scale table) and writes one (round 5: `chscale_fwd_kernel`, the calibration kernel
of rounds 1-4, is folded into w_conv and no longer in the trace).  The script FAILS when the calibration
kernel is missing or the corrected read figure is off by more than 10 %.
"""
import json
import sqlite3
import sys

fetch_db, write_db, out_md, out_json = sys.argv[1:5]
DOM = sys.argv[5] if len(sys.argv) > 5 else "conv_h2w2_kernel<7"


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def per_dispatch(path, counter):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select dispatch_id, kernel_name, grid_size, sum(value) from counters_collection where counter_name=? "
                       "group by dispatch_id order by dispatch_id", (counter,)).fetchall()
    return [(short(n), g, v) for _, n, g, v in rows]


F, W = per_dispatch(fetch_db, "FETCH_SIZE"), per_dispatch(write_db, "WRITE_SIZE")
assert [x[:2] for x in F] == [x[:2] for x in W], "the two passes must launch the same kernel sequence"
stats = {}
for (n, g, f), (_, _, w) in zip(F, W):
    stats.setdefault(n, []).append((f, w))
# calibration: chscale_bwd_kernel over the [256, 384, 1250] encoder output (config 2) reads 2 x 491,520,000 bytes and writes 491,520,000
CAL, CAL_BYTES = "chscale_bwd_kernel", float(sys.argv[6]) if len(sys.argv) > 6 else 256 * 384 * 1250 * 4.0
cal = [(f, w) for (n, g, f), (_, _, w) in zip(F, W) if n == CAL]
if not cal:
    sys.exit(f"pmc_table: calibration kernel {CAL} not in the trace -- refusing to emit an uncalibrated table")
cal_ratio = (sum(f for f, _ in cal) / len(cal)) * 1024 / (2 * CAL_BYTES)
cal_w = (sum(w for _, w in cal) / len(cal)) * 1024 / CAL_BYTES
if not (0.40 < cal_ratio < 0.62) or not (0.85 < cal_w < 1.15):
    sys.exit(f"pmc_table: calibration off: FETCH_SIZE x 1024 = {cal_ratio:.3f} of the known read bytes (expected 0.5), "
             f"WRITE_SIZE x 1024 = {cal_w:.3f} of the known write bytes (expected 1.0)")
# dominant kernel (bench.py's `roofline`): the kernel with the largest summed duration in the step -- since round 4 the K = 7
# weight gradient conv_h2w2_kernel<7, 0, ...>; every launch of a kernel name is averaged
k7 = [(f, w) for (n, g, f), (_, _, w) in zip(F, W) if n.startswith(DOM)]
if not k7:
    sys.exit(f"pmc_table: no launch of {DOM} in the trace")
all_bytes = sum((2 * f + w) * 1024 for f, w in k7) / len(k7)
with open(out_md, "w") as fh:
    fh.write("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 1, config 2 "
             "(B=256,V=3,L=5000)\n\n")
    fh.write("Counter unit: KB per dispatch.  HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 -- gfx950's FETCH_SIZE counts\n"
             "half of the bytes read (MI355X guide).  Calibration in this run: `chscale_bwd_kernel` reads 2 x and writes\n"
             f"{CAL_BYTES:,.0f} bytes per launch; FETCH_SIZE x 1024 = {cal_ratio:.3f} of the read, WRITE_SIZE x 1024 = {cal_w:.3f} of "
             "the write.  Fabric-side counters: hits in the memory-side\n"
             "cache (MALL) are counted like HBM reads, so these are upper bounds on DRAM traffic.\n\n")
    fh.write(f"Dominant kernel `{DOM}...>`: {len(k7)} launches, {all_bytes/1e9:.3f} GB per launch on average "
             "(algorithmic: one read of each operand row, 0.984 GB for the K = 7 weight gradient -- the two 64-channel "
             "column tiles of a (group, split) both read the 128 gradient rows, through the L2 of one XCD).\n\n")
    fh.write("| kernel | launches | FETCH_SIZE avg (min..max) KB | WRITE_SIZE avg (min..max) KB | corrected bytes/launch (GB) |\n"
             "|---|---:|---:|---:|---:|\n")
    order = sorted(stats.items(), key=lambda kv: -sum((2 * f + w) for f, w in kv[1]))
    for n, v in order:
        fs, ws = [f for f, _ in v], [w for _, w in v]
        gb = sum((2 * f + w) * 1024 for f, w in v) / len(v) / 1e9
        fh.write(f"| `{n}` | {len(v)} | {sum(fs)/len(fs):.0f} ({min(fs):.0f}..{max(fs):.0f}) | "
                 f"{sum(ws)/len(ws):.0f} ({min(ws):.0f}..{max(ws):.0f}) | {gb:.3f} |\n")
json.dump({"by_kernel": {n: int(sum((2 * f + w) * 1024 for f, w in v) / len(v)) for n, v in stats.items()},
           "dominant": DOM, "dominant_bytes_per_launch": int(all_bytes),
           "source": out_md + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
           "note": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024, mean over the launches of a kernel name"},
          open(out_json, "w"), indent=1)
print(open(out_md).read()[:1800])
