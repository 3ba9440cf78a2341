#!/usr/bin/env python3
"""usage: bh_report.py <backhalf_sq.json> [kernel substring ...] -- the stall picture of tools/prof_backhalf.sh's counters, per kernel."""
import json
import sys

d = json.load(open(sys.argv[1]))
subs = sys.argv[2:] or ["reduce_", "km_write"]
for k, a in d.items():
    if not any(s in k for s in subs) or "SQ_WAVE_CYCLES" not in a:
        continue
    wc = a["SQ_WAVE_CYCLES"]
    g = lambda c: a.get(c, 0.0)
    print(k, "launches", a["launches"], "waves", g("SQ_WAVES"), "vgpr", a.get("VGPR_Count"))
    print("  of wave cycles: WAIT_ANY %.1f%%  WAIT_INST_ANY %.1f%% (LDS %.1f%%)  ACTIVE %.1f%% (VALU %.1f%% LDS %.1f%% SCA %.1f%%)" % tuple(
        100 * g(c) / wc for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA")))
    w = max(1.0, g("SQ_WAVES"))
    print("  per wave: VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.1f VMEM_WR %.1f   wave life %.0f quad-cycles" % (
        g("SQ_INSTS_VALU") / w, g("SQ_INSTS_SALU") / w, g("SQ_INSTS_LDS") / w, g("SQ_INSTS_VMEM_RD") / w, g("SQ_INSTS_VMEM_WR") / w, wc / w))
    print("  LDS per CU: idx_active %.3g cycles, bank conflicts %.0f%%, data fifo full %.3g, cmd fifo full %.3g" % (
        g("SQ_LDS_IDX_ACTIVE") / 256, 100 * g("SQ_LDS_BANK_CONFLICT") / max(1, g("SQ_LDS_IDX_ACTIVE")), g("SQ_LDS_DATA_FIFO_FULL") / 256, g("SQ_LDS_CMD_FIFO_FULL") / 256))
    print("  VMEM in flight per wave-cycle %.2f; TCP per CU: pending stall %.3g, TA addr stalled by TC %.3g, read req %.3g (lat %.0f), write req %.3g" % (
        g("SQ_INST_LEVEL_VMEM") / wc, g("TCP_PENDING_STALL_CYCLES_sum") / 256, g("TA_ADDR_STALLED_BY_TC_CYCLES_sum") / 256, g("TCP_TCC_READ_REQ_sum") / 256,
        g("TCP_TCC_READ_REQ_LATENCY_sum") / max(1, g("TCP_TCC_READ_REQ_sum")), g("TCP_TCC_WRITE_REQ_sum") / 256))
    print("  TCC: req %.3g hit %.0f%%  EA rd %.3g wr %.3g wr_stall %.3g" % (g("TCC_REQ_sum"), 100 * g("TCC_HIT_sum") / max(1, g("TCC_REQ_sum")), g("TCC_EA0_RDREQ_sum"), g("TCC_EA0_WRREQ_sum"), g("TCC_EA0_WRREQ_STALL_sum")))
