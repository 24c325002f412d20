"""profiles/sq_k7.json from an SQ-counter summary (tools/pmc_sq.sh -> sq.md): per kernel the average duration, the effective clock
(GRBM_GUI_ACTIVE per XCD / duration) and the matrix-pipe occupancy -- what bench.py replays into roofline.effective_clock_GHz /
pipe_busy.   usage: python tools/sq_to_json.py gpurun_out/r05/sq/sq.md profiles/sq_k7.json "profiles/r05_sq_counters_k7.md (...)" """
import json, re, sys

src, dst, label = sys.argv[1], sys.argv[2], sys.argv[3]
out, cur = {}, None
for line in open(src):
    m = re.match(r"^(\S.*?)\s+avg_dur_us=([\d.]+)", line)
    if m:
        cur = out.setdefault(m.group(1), {"avg_dur_us": float(m.group(2))})
        continue
    m = re.match(r"^\s+(GRBM_GUI_ACTIVE|SQ_VALU_MFMA_BUSY_CYCLES)\s+([\d.e+]+)", line)
    if m and cur is not None:
        cur[m.group(1)] = float(m.group(2))
    m = re.match(r"^\s+matrix-pipe occupancy\s+([\d.]+)", line)
    if m and cur is not None:
        cur["pipe_busy"] = float(m.group(1))
res = {}
for k, v in out.items():
    if "GRBM_GUI_ACTIVE" in v and v["avg_dur_us"] > 0:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles per XCD / duration = the clock the launch ran at
        res[k] = {"avg_dur_us": v["avg_dur_us"], "effective_clock_GHz": round(v["GRBM_GUI_ACTIVE"] / 8 / (v["avg_dur_us"] * 1e3), 3),
                  "pipe_busy": v.get("pipe_busy", 0.0)}
json.dump({"source": label, "by_kernel": res}, open(dst, "w"), indent=1)
print(len(res), "kernels ->", dst)
