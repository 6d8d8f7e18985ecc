#!/usr/bin/env python3
"""Where k_decode_chain's wave-instructions go, per phase, for one decoded llama-2-7B W2 token.

Static model: instructions per basic block of the benched instantiation k_decode_chain<2, true, true, 0, false> (hipcc -S
-gline-tables-only, tools/isa_phase_count.py --bb: blocks classified by the source lines their instructions come from) x the trip
counts the op descriptors imply (waves per quad as tmac_chain_host.cpp's chain_pick_wpq chooses them, 256 workgroups of 12 waves,
contiguous balanced quad ranges).  Exec-masked instructions count as issued (SQ_INSTS_VALU counts them too); whole-wave skips through s_cbranch_execz are modelled (the
round slots of the LUT build in which a wave has no pairs).  Cross-check: SQ_INSTS_VALU / SQ_INSTS_SALU of the same launch: 2.08e8 / 1.29e8
per token in round 4 (profiles/r04b_rocprof_summary.txt), re-measured at this commit in profiles/r05_rocprof_summary.txt; the SALU of the
poll loops' s_sleep spins is not in the model.

usage: chain_valu_model.py            prints the table (markdown) for the llama-2-7B W2 decode chain
"""
NWV, GRID, RING = 12, 256, 4

# (VALU, SALU, MFMA) per execution of a phase's code, from the listing of tmac_chain.hip at this commit (block labels in comments)
PH = {
    "op entry: descriptor fields (LDS reads + readfirstlane), role of the wave":         (29, 43, 0),    # BB5_7..13
    "hand-off: poll set-up, one poll round (loads, tag compares, ballot), exit":        (19 + 26 + 14, 13 + 40 + 25, 0),   # BB5_44/47, 60..66, 82/85
    "issue of one item: quad -> matrix / offsets, lane's scale address, 3 loads":        (16, 37, 0),    # BB5_15/20.. + 25/27 (refills: 174/176)
    "LUT build, per wave and round slot (3 unrolled) WITHOUT pairs: the test, skipped by s_cbranch_execz": (4, 8, 0),
    "LUT build, per wave and ACTIVE round (64 pairs = 128 tables): unpack, |x| sums, group max, /127, 1/scale, 2 x q_table8, bias sums, LDS stores": (42 + 73, 17 + 6, 0),     # BB5_124 + 128 + 131
    "zero tables of the padded step, barrier, lookup-loop set-up":                       (8 + 2 + 12, 14 + 26 + 11, 0),   # BB5_151..167
    "item: 4 table reads, 32 lookups x (and, perm, shift, and_or, perm), 8 MFMA adds":   (105, 17, 8),   # BB5_169 (199, 231, 262)
    "item: scale chain of the lane's two act groups (cvt, 2-3 fma each)":                (13, 6, 0),     # head of BB5_178
    "finish of a workgroup iteration, every wave: DPP / permlane reduce, partial to LDS": (31 + 3, 6 + 7, 0),   # rest of BB5_178 + 184
    "finish, wave 0 only: combine split quads, fp16 round, store, publish granules":     (2 + 6 + 15 + 1, 3 + 6 + 6 + 5, 0),   # BB5_188..196
    "op exit":                                                                           (3 + 3 + 5, 14 + 12 + 17, 0),   # BB5_303/305/328
}


def pick_wpq(total_q, nst, grid=GRID):
    best, best_cost = 1, 1 << 60
    for wpq in (1, 2, 3, 4):
        if NWV % wpq or (wpq > 1 and wpq > nst):
            continue
        ipi = NWV // wpq
        cnt = -(-total_q // grid)
        cost = -(-cnt // ipi) * -(-nst // wpq)
        if cost < best_cost:
            best, best_cost = wpq, cost
    return best


def op_counts(K, rows):
    """per workgroup of the busiest kind (cnt = ceil): items, active build rounds, workgroup iterations"""
    nu, total_q = K // 32, sum(rows) // 4
    nst = -(-nu // 64)
    wpq = pick_wpq(total_q, nst)
    ipi = NWV // wpq
    cnt = total_q / GRID                       # average quads per workgroup
    items = cnt * nst                          # (quad, step) items per workgroup, spread over the waves
    iters = -(-int(-(-total_q // GRID)) // ipi)
    P = K // 8
    active_rounds = -(-P // 64)                # 64-pair blocks = (wave, round) slots that build tables
    return dict(items=items, iters=iters, active=active_rounds, wpq=wpq, nst=nst)


LLAMA = [("q/k/v", 4096, [4096] * 3), ("o", 4096, [4096]), ("gate/up", 4096, [11008] * 2), ("down", 11008, [4096])]
LAYERS = 32

tot = {k: [0.0, 0.0, 0.0] for k in PH}
keys = list(PH)
for name, K, rows in LLAMA:
    c = op_counts(K, rows)
    per_cu = {
        keys[0]: NWV, keys[1]: NWV, keys[2]: c["items"], keys[3]: 3 * NWV - c["active"], keys[4]: c["active"], keys[5]: NWV,
        keys[6]: c["items"], keys[7]: c["items"], keys[8]: NWV * c["iters"], keys[9]: c["iters"], keys[10]: NWV,
    }
    for k, n in per_cu.items():
        for j in range(3):
            tot[k][j] += n * PH[k][j] * LAYERS * GRID

sv = sum(v[0] for v in tot.values()); ss = sum(v[1] for v in tot.values()); sm = sum(v[2] for v in tot.values())
print("| phase (k_decode_chain<2,true,true,0,false>, llama-2-7B W2 token: 128 ops, 256 workgroups x 12 waves) | VALU | % | SALU | % | on the op's critical path? |")
print("|---|---|---|---|---|---|")
crit = {0: "no (waves enter an op while the slowest producer still publishes)", 1: "the poll's round trip is; its instructions are not",
        2: "no (issue is bound by the CU's address pipe, 64 B/clk)", 3: "yes", 4: "yes", 5: "yes (short)", 6: "yes", 7: "yes", 8: "yes", 9: "yes (wave 0)", 10: "no"}
for i, k in enumerate(keys):
    v = tot[k]
    print(f"| {k} | {v[0]:.3g} | {100 * v[0] / sv:.0f} | {v[1]:.3g} | {100 * v[1] / ss:.0f} | {crit[i]} |")
print(f"| **total per token (model)** | **{sv:.3g}** | 100 | **{ss:.3g}** | 100 | MFMA {sm:.3g} |")
print()
print("per item: %d VALU + %d MFMA + %d SALU executed (lookups 88 = 32 x 11/4, table / scale reads 8 LDS); per 2 KB of weights" % (105 + 13 + 16, 8, 17 + 6 + 37))
