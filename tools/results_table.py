#!/usr/bin/env python3
"""DESIGN.md section 5's table of numbers, generated from the committed evidence (profiles/r05_bench_*.json, r05_bench_stream.txt,
r05_rocprof_summary.txt) so that the document and the files cannot drift apart.  usage: tools/results_table.py > table.md"""
import json, os, re
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
def J(w): return json.load(open(os.path.join(R, f"r05_bench_{w}.json")))
w2, w4, bn = J("llama2-7b-w2"), J("llama2-7b-w4"), J("bitnet-3b")
ind, dec, s20 = J("independent_pattern"), J("decoder_pattern"), J("default_steps20")
p2, p4, pb = J("llama2-7b-w2-prefill"), J("llama2-7b-w4-prefill"), J("bitnet-3b-prefill")
summ = open(os.path.join(R, "r05_rocprof_summary.txt")).read()
def avg(pat):
    m = re.search(pat + r".*?calls\s+(\d+)\s+avg\s+([\d.]+)", summ)
    return (int(m.group(1)), float(m.group(2))) if m else (0, 0.0)
def fetch(pat):
    m = re.search(pat + r".*?FETCH_SIZE [\d.]+ KiB -> x2 = [\d.]+ MB \((\d+) B\)", summ)
    return int(m.group(1)) if m else 0
bs = {}
for l in open(os.path.join(R, "r05_bench_stream.txt")):
    m = re.match(r"(\S+) W(\d) (stream|chain)\s+wpq=\d:\s+([\d.]+) us/call.*?frac ([\d.]+)", l)
    if m: bs[(m.group(1), int(m.group(2)), m.group(3))] = (float(m.group(4)), float(m.group(5)))
r2 = w2["roofline"]
nc, ac = avg(r"k_decode_chain<2, true, true, 0, false>")
ns, as_ = avg(r"k_gemv_stream<2, true, true, 2, 8, 0>")
nq, aq = avg(r"k_gemv_quad<2, true, 0, 1, 6, 768, 3")
rows = [
 ("llama-2-7B W2 decode chain (`value`; rocprof: `k_decode_chain` avg %.1f µs over %d calls; %.3f ms with `--steps 20`)" % (ac, nc, s20["ms_per_step"]), "%.3f ms" % w2["ms_per_step"], "%.0f GB/s" % w2["value"], "%.3f" % r2["frac"]),
 ("… W4 / BitNet-3B chains", "%.3f / %.3f ms" % (w4["ms_per_step"], bn["ms_per_step"]), "%.0f / %.0f GB/s" % (w4["value"], bn["value"]), "%.3f / %.3f" % (w4["roofline"]["frac"], bn["roofline"]["frac"])),
 ("headline GEMV 4096 × 11008, stand-alone launches (`k_gemv_quad` avg %.2f µs under rocprof)" % aq, "%.2f µs" % r2["headline_gemv"]["us"], "%.0f GB/s" % r2["headline_gemv"]["GBps"], "%.3f" % r2["headline_gemv"]["frac"]),
 ("headline GEMV, stream mode (`tools/bench_stream.py` / `roofline.stream_core` of the default line)", "%.2f / %.2f µs" % (bs[("4096x11008x1", 2, "stream")][0], r2["stream_core"]["us_per_gemv"]), "", "%.2f / %.2f" % (bs[("4096x11008x1", 2, "stream")][1], r2["stream_core"]["frac"])),
 ("gate/up, q/k/v, o in stream mode (`bench_stream.py`; `roofline.stream_by_shape` of the default line: %s)" % " / ".join("%.2f" % r2["stream_by_shape"][k]["frac"] for k in ("gate_up", "qkv", "o")),
  "%.2f / %.2f / %.2f µs" % (bs[("11008x4096x2", 2, "stream")][0], bs[("4096x4096x3", 2, "stream")][0], bs[("4096x4096x1", 2, "stream")][0]), "",
  "%.2f / %.2f / %.2f" % (bs[("11008x4096x2", 2, "stream")][1], bs[("4096x4096x3", 2, "stream")][1], bs[("4096x4096x1", 2, "stream")][1])),
 ("the same four shapes through `k_decode_chain` (TMAC_CHAIN_STREAM=0)", " / ".join("%.2f" % bs[(k, 2, "chain")][0] for k in ("4096x11008x1", "11008x4096x2", "4096x4096x3", "4096x4096x1")) + " µs", "", " / ".join("%.2f" % bs[(k, 2, "chain")][1] for k in ("4096x11008x1", "11008x4096x2", "4096x4096x3", "4096x4096x1"))),
 ("W3 / W4 / W1 headline GEMV in stream mode", "%.2f / %.2f / %.2f µs" % (bs[("4096x11008x1", 3, "stream")][0], bs[("4096x11008x1", 4, "stream")][0], bs[("4096x11008x1", 1, "stream")][0]), "", "%.2f / %.2f / %.2f" % (bs[("4096x11008x1", 3, "stream")][1], bs[("4096x11008x1", 4, "stream")][1], bs[("4096x11008x1", 1, "stream")][1])),
 ("token as independent calls, stream mode (`--pattern independent`; `k_gemv_stream` avg %.0f µs under rocprof; %.3f ms in the default line)" % (as_, r2["independent_pattern"]["ms_per_token"]), "%.3f ms" % ind["ms_per_step"], "%.0f GB/s" % ind["value"], "%.3f" % ind["roofline"]["frac"]),
 ("… W4 / BitNet-3B (`roofline.independent_pattern` of their lines)", "%.3f / %.3f ms" % (w4["roofline"]["independent_pattern"]["ms_per_token"], bn["roofline"]["independent_pattern"]["ms_per_token"]), "", "%.2f / %.2f" % (w4["roofline"]["independent_pattern"]["frac"], bn["roofline"]["independent_pattern"]["frac"])),
 ("BitNet-3B 3200 × 8640 GEMV, stream mode (`roofline.stream_core`)", "%.2f µs" % bn["roofline"]["stream_core"]["us_per_gemv"], "%.0f GB/s" % bn["roofline"]["stream_core"]["GBps"], "%.2f" % bn["roofline"]["stream_core"]["frac"]),
 ("decoder pattern (`--pattern decoder` / in the default line)", "%.3f / %.3f ms" % (dec["ms_per_step"], r2["decoder_pattern"]["ms_per_token"]), "", "%.3f" % dec["roofline"]["frac"]),
 ("prefill 256 tokens W2 / W4 / BitNet", "%.2f / %.2f / %.2f ms" % (p2["ms_per_step"], p4["ms_per_step"], pb["ms_per_step"]), "%.0f / %.0f / %.0f tok/s" % (p2["value"], p4["value"], pb["value"]), "%.3f / %.3f / %.3f (mfma, one-hot operand)" % (p2["roofline"]["frac"], p4["roofline"]["frac"], pb["roofline"]["frac"])),
 ("… dense fp16 `torch.matmul` of the same shapes, graph-replayed, same run", "%.2f / %.2f / %.2f ms" % tuple(x["roofline"]["dense_fp16_baseline"]["ms_per_step"] for x in (p2, p4, pb)), "", "this is %.2f × / %.2f × / %.2f × its tokens/s" % tuple(x["roofline"]["dense_fp16_baseline"]["this_over_dense"] for x in (p2, p4, pb))),
 ("… in the default decode line (`prefill_twin`, 30 steps; `other_decode_chains` W4 / BitNet)", "%.2f ms; %.3f / %.3f ms" % (w2["prefill_twin"]["ms_per_step"], w2["other_decode_chains"]["llama2-7b-w4"]["ms_per_step"], w2["other_decode_chains"]["bitnet-3b"]["ms_per_step"]), "", ""),
 ("reference CPU kernels (`oracle/_ref`), best of %s host threads = %d" % (" / ".join(w2["cpu_baseline"]["by_threads_GBps"].keys()), w2["cpu_baseline"]["cores"]), "", "%.0f GB/s (W2 decode), %.0f tok/s (W2 prefill)" % (w2["cpu_baseline"]["value"], p2["cpu_baseline"]["value"]), ""),
]
print("| round-5 numbers (final evidence run, one box; generated by `tools/results_table.py` from `profiles/r05_bench_*.json`, `r05_bench_stream.txt`, `r05_rocprof_summary.txt`) | ms / µs | GB/s or tok/s | frac |")
print("|---|---|---|---|")
for r in rows: print("| " + " | ".join(r) + " |")
alg = w2["config"]["algorithmic_bytes_per_step"]
fc, fs, fl = fetch(r"fetch_chain[\s\S]*?k_decode_chain<2, true, true, 0, false>"), fetch(r"k_gemv_stream<2, true, true, 2, 8, 0>\s+launches"), fetch(r"k_lut_images\s+launches")
print("HBM traffic (`FETCH_SIZE` × 2): chain %s B = %.3f × algorithmic; `k_gemv_stream` %s B (+ %.2f MB `k_lut_images`) = %.3f ×." % (f"{fc:,}", fc / alg, f"{fs:,}", fl / 1e6, (fs + fl) / alg))
