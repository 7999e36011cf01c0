#!/usr/bin/env python3
"""Derived-metrics header for a gemm_pmc_counters file (the `== kernel` / `pass N {...}` lines tools/pmc_bench.sh prints): matrix-pipe
utilisation, L2 hit rate, mean L2 read latency, LDS bank-conflict share, wave cycles parked -- what bench.py's roofline.mfma_util /
l2_hit are read from (bench.pmc_derived).  usage: pmc_counters_report.py RAW.txt TAG > profiles/rNN_gemm_pmc_counters.txt"""
import ast, sys
raw, tag = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
kern, cur, lines = {}, None, []
for ln in open(raw):
    ln = ln.rstrip("\n")
    if ln.startswith("== "):
        cur = kern.setdefault(ln[3:].strip(), {}); lines.append(ln)
    elif ln.startswith("pass ") and cur is not None and "{" in ln:
        cur.update(ast.literal_eval(ln[ln.index("{"):ln.rindex("}") + 1])); lines.append(ln)
print("# rocprofv3 --pmc passes (tools/pmc_bench.sh: three separate passes per kernel family over `bench.py --steps 1 --layers 3`, config-2")
print("# shapes), per-launch averages%s.  Kernel names are matched by substring: gemm_bf16_w16_kernel = QKV (16-bit epilogue) and fc1" % (", " + tag if tag else ""))
print("# (bias + GELU) launches together, gemm_bf16_pp_kernel = out-projection and fc2 (fp32 residual epilogue), attention_kernel = attention_kernel<18>.")
print("# Derived: mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)  (busy matrix-pipe cycles per SIMD / kernel cycles;")
print("# profiled passes run at lower clocks); l2_hit = TCC_HIT / (TCC_HIT + TCC_MISS); L2 read latency = TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ;")
print("# LDS bank-conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked in s_waitcnt / barriers).")
for k, c in kern.items():
    mf = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0) if c.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in c else float("nan")
    hit = c["TCC_HIT"] / (c["TCC_HIT"] + c["TCC_MISS"]) if c.get("TCC_HIT") is not None and c.get("TCC_HIT", 0) + c.get("TCC_MISS", 0) else float("nan")
    lat = c["TCP_TCC_READ_REQ_LATENCY"] / c["TCP_TCC_READ_REQ"] if c.get("TCP_TCC_READ_REQ") else float("nan")
    bc = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else float("nan")
    wt = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else float("nan")
    print("#   %-24s matrix-pipe utilisation %.2f, L2 hit rate %.2f, mean L2 read latency %.0f cycles, LDS bank-conflict share %.3f, wave cycles waiting %.2f"
          % (k, mf, hit, lat, bc, wt))
print("\n".join(lines))
