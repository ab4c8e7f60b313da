#!/usr/bin/env python3
"""profiles/rNN_bench_process_breakdown.txt + profiles/rNN_pmc_bench.json -> profiles/rNN_kernel_roofline.md: per kernel
class, time per process(), algorithmic work, achieved fp32-equivalent TFLOP/s against the kernel's own MFMA ceiling
(dense fp16 2500 TF / passes), matrix-pipe occupancy and bytes past L2 from the in-situ counter passes.
usage: python tools/kernel_roofline_md.py r04 [r03]      (second tag: previous round, for the delta column)"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
prev = sys.argv[2] if len(sys.argv) > 2 else None
B, T = 4096, 20


def breakdown(t):
    rows = {}
    head = []
    for l in open(os.path.join(ROOT, "profiles", "%s_bench_process_breakdown.txt" % t)):
        if l.startswith("#"):
            head.append(l[2:].strip())
            continue
        m = re.match(r"(.+?)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", l)
        if m:
            rows[m.group(1).strip()] = dict(ms=float(m.group(2)), calls=float(m.group(3)), avg=float(m.group(4)))
    return head, rows


head, br = breakdown(tag)
pbr = breakdown(prev)[1] if prev else {}
pmc = json.load(open(os.path.join(ROOT, "profiles", "%s_pmc_bench.json" % tag)))["kernels"]


def pick(rows, *subs):
    out = [(k, v) for k, v in rows.items() if all(s in k for s in subs)]
    return sum(v["ms"] for _, v in out), sum(v["calls"] for _, v in out)


def pm(name, key):
    for k, v in pmc.items():
        if k.startswith(name):
            return v.get(key)
    return None


rowsB, rows2 = T * B, 2 * T * B                          # 81,920 / 163,840
frames_bwd = rowsB + rows2 + 3 * B                       # base + batched replay pass + reward prediction
frames_fwd = 21 * B + 2 * B + rows2 + 3 * B              # 20 rollout steps + bootstrap, the pass's bootstrap pair, the pass, rp
G = 1e9
nt_flop = 2.0 * (rowsB + rows2) * 256 * 2592 * 2 + 2.0 * (rowsB + rows2) * 256 * 1024 + 2.0 * rowsB * 2592 * 256 * 2   # fc fwd(aux only at 128^2) ...
# 128 x 128 NT launches of one call: fc fwd of the replay pass (163,840 rows), fc dgrad x2 (81,920 + 163,840), LSTM-input dgrad x2,
# pc_fc1 fwd + dgrad (81,920 rows each)
nt_flop = 2.0 * rows2 * 256 * 2592 + 2.0 * (rowsB + rows2) * 2592 * 256 + 2.0 * (rowsB + rows2) * 256 * 1024 + 2.0 * rowsB * 2592 * 256 * 2
tn_flop = 2.0 * (rowsB + rows2) * 2592 * 256 + 2.0 * (rowsB + rows2) * 256 * 1024 * 2 + 2.0 * rowsB * 256 * 2592
lstm_flop = 2.0 * (21 * B + 21 * 2 * B) * 1024 * (288 + 256)
bptt_flop = 2.0 * (19 * B + 19 * 2 * B) * 256 * 1024
fc_flop = 2.0 * 21 * B * 256 * 2592
items = [
    ("encoder_bwd (3: base, batched replay pass, reward prediction)", ("encoder_bwd",), frames_bwd * 5.11e6, 992, "encoder_bwd_roles_kernel",
     "%d frames x 5.11 MFLOP, x 57.1 KB" % frames_bwd),
    ("gemm_split_nt 128 x 128 (8: fc fwd of the pass, fc dgrad x2, LSTM-input dgrad x2, pc_fc1 fwd + dgrad)", ("gemm_split_nt_kernel<128, 128, true, false, 0",), nt_flop, 833,
     "gemm_split_nt_kernel<128, 128, true, false, 0", "%.0f GFLOP" % (nt_flop / G)),
    ("encoder_fwd (24)", ("encoder_fwd",), frames_fwd * 3.78e6, 1103, "encoder_fwd_kernel", "%d frames x 3.78 MFLOP" % frames_fwd),
    ("gemm_split_tn (8: wgrad + bias grad)", ("gemm_split_tn",), tn_flop, 833, "gemm_split_tn_kernel", "%.0f GFLOP" % (tn_flop / G)),
    ("LSTM step, [x | h] . kernel + gates (21 x 4096 + 21 x 8192 rows)", ("gemm_split_nt_kernel<128, 128, true, false, 1",), lstm_flop, 833,
     "gemm_split_nt_kernel<128, 128, true, false, 1, 1", "%.0f GFLOP" % (lstm_flop / G)),
    ("BPTT step, dgrad + earlier step's gate backward (19 x 4096 + 19 x 8192 rows)", ("gemm_split_nt_kernel<64, 64, true, true, 2",), bptt_flop, 833,
     "gemm_split_nt_kernel<64, 64, true, true, 2, 2", "%.0f GFLOP + gate tensors" % (bptt_flop / G)),
    ("fc 2592->256 of a rollout step (21 x 4096 rows)", ("gemm_split_nt_kernel<64, 64, true, true, 0, 4",), fc_flop, 833,
     "gemm_split_nt_kernel<64, 64, true, true, 0, 4", "%.0f GFLOP" % (fc_flop / G)),
] + ([
    # round 4 (late): the training pass of the head is one launch; the forward kernel is left with the bootstrap Q-max
    ("pc_deconv_train (loss + backward in one launch; before: pc_deconv_fwd training launch + pc_deconv_bwd)", ("pc_deconv_train",),
     rowsB * 1.24e6, 833, "pc_deconv_train_kernel", "81,920 frames x 1.24 MFLOP, x 22.3 KB"),
    ("pc_deconv_fwd (bootstrap Q-max)", ("pc_deconv_fwd",), B * 0.41e6, 833, "pc_deconv_fwd_kernel", "4,096 frames x 0.41 MFLOP"),
] if pick(br, "pc_deconv_train")[0] else [
    ("pc_deconv_bwd", ("pc_deconv_bwd",), rowsB * 0.83e6, 833, "pc_deconv_bwd_kernel", "81,920 frames x 0.83 MFLOP"),
    ("pc_deconv_fwd (2)", ("pc_deconv_fwd",), (rowsB + B) * 0.41e6, 833, "pc_deconv_fwd_kernel", "86,016 frames x 0.41 MFLOP"),
]) + [
    ("maze_step + policy head (20; fused since round 4)", ("maze_step_kernel",), None, None, "maze_step_kernel", "20 x 4096 actors x 22.8 KB"),
]
out = ["# Per-kernel achieved rates inside `bench.py` (MI355X, round %s)" % tag[1:].lstrip("0"), "",
       "Generated by `tools/kernel_roofline_md.py %s` from `profiles/%s_bench_process_breakdown.txt` (%s; %s) and the in-situ counter "
       "passes `profiles/%s_pmc_bench.json`.  GFLOP are fp32-equivalent (a product counts once, however many 16-bit MFMA passes "
       "implement it); ceilings: dense fp16 MFMA 2500 TFLOP/s nominal over the passes a product needs (3: fp16 hi + lo x hi + lo; "
       "2 where one operand is a uint8 pixel), HBM 8 TB/s nominal (a streaming kernel reaches ~6.2)." % (tag, tag, head[0], head[2], tag), "",
       "| kernel | ms / process()%s | work per process() | TFLOP/s (fp32-eq.) | MFMA ceiling: fraction | matrix pipe busy | bytes past L2 per launch -> TB/s |" % (
           " (%s)" % prev if prev else ""), "|---|---|---|---|---|---|---|"]
listed = 0.0
for name, subs, flop, ceil, pk, work in items:
    ms, calls = pick(br, *subs)
    pms = pick(pbr, *subs)[0] if prev else None
    if name.startswith("maze_step") and prev:
        pms += pick(pbr, "policy_step")[0]
    if name.startswith("pc_deconv_train") and prev:          # the previous round's two launches of the same work
        pms = pick(pbr, "pc_deconv_bwd")[0] + pick(pbr, "pc_deconv_fwd")[0] * (rowsB / float(rowsB + B))
    listed += ms
    tf = flop / (ms * 1e-3) / 1e12 if flop and ms else None
    busy, hb = pm(pk, "mfma_busy_frac"), pm(pk, "hbm_bytes_per_launch")
    avg_s = ms * 1e-3 / calls if calls else 0
    out.append("| %s | %.2f%s | %s | %s | %s | %s | %s |" % (
        name, ms, " (%.2f)" % pms if pms else "", work, "%.0f" % tf if tf else "-",
        "%d: %.2f" % (ceil, tf / ceil) if tf else "-", "%.2f" % busy if busy is not None else "-",
        "%.2f GB -> %.1f" % (hb / 1e9, hb / avg_s / 1e12) if hb and avg_s else "-"))
total = sum(v["ms"] for v in br.values())
out.append("| everything else (heads, losses, scans, sampling, fills, weight shadows, RMSProp) | %.2f | | | | | |" % (total - listed))
out += ["", "Sum of the kernel times: %.2f ms per process() (%s)." % (total, head[2])]
tot_bytes = 0.0
for k, v in pmc.items():
    if v.get("hbm_bytes_per_launch") and v.get("launches_per_call"):
        tot_bytes += v["hbm_bytes_per_launch"] * v["launches_per_call"]
out.append("Bytes past L2, all kernels: %.1f GB per call = %.2f TB/s averaged over the call." % (tot_bytes / 1e9, tot_bytes / (total * 1e-3) / 1e12))
open(os.path.join(ROOT, "profiles", "%s_kernel_roofline.md" % tag), "w").write("\n".join(out) + "\n")
print("\n".join(out))
