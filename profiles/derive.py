#!/usr/bin/env python3
"""profiles/rNN_*.txt (tools/profile_round6.sh: kernel trace + one PMC group per pass, per (kernel, grid)) -> the derived figures the bench
line and DESIGN.md quote, per (kernel, grid): ms per launch (trace), sustained clock (GRBM_GUI_ACTIVE / 8 XCDs / duration), matrix-pipe busy
(SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs)), VALU issue utilisation (SQ_INSTS_VALU x 4 cycles / (cycles x 1024)), VALU busy
(SQ_ACTIVE_INST_VALU x 4 / (cycles x 1024)), VALU per MFMA, LDS bank-conflict cycles per launch and per CU-cycle, HBM bytes per launch
(FETCH_SIZE KiB x 2 on gfx950 + WRITE_SIZE KiB, MI355X_MICROARCH.md) and GB/s, L2 hit rate.
usage: python profiles/derive.py profiles/r06_final.txt [--min-ms 0.2] > profiles/r06_final_derived.json"""
import json
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("pr::", "").replace("void ", "").strip()
    m = re.match(r"([A-Za-z_]\w*(?:<[^()]*?>)?)", name)
    return m.group(1) if m else name


def main():
    path = sys.argv[1]
    min_ms = float(sys.argv[sys.argv.index("--min-ms") + 1]) if "--min-ms" in sys.argv else 0.2
    trace, clusters, pmc, order = {}, {}, {}, []
    mode = None
    for l in open(path):
        if l.startswith("-- kernel trace"):
            mode = "trace"; continue
        if l.startswith("-- duration clusters"):
            mode = "clusters"; continue
        if l.startswith("-- PMC"):
            mode = "pmc"; continue
        if mode == "trace":
            m = re.match(r"^(.*?)\s+n=(\d+)\s+avg=\s*([\d.]+) min=\s*([\d.]+) max=\s*([\d.]+) total=\s*([\d.]+) vgpr=(\d+) agpr=(\d+) lds=(\d+) grid=(\d+)", l)
            if m and "pr::" in m.group(1):
                order.append((short(m.group(1)), int(m.group(10))))
                trace[(short(m.group(1)), int(m.group(10)))] = {"n": int(m.group(2)), "ms_avg": float(m.group(3)), "ms_min": float(m.group(4)), "ms_max": float(m.group(5)),
                                                               "vgpr": int(m.group(7)), "lds": int(m.group(9))}
        elif mode == "clusters":
            # launches of one (kernel, grid) that differ widely (one channel | both channels; gated launches that leave at once): the trace's
            # duration clusters, ascending - matched by rank with the PMC passes' clusters of the same (kernel, grid).  The cluster lines follow
            # the trace rows' order; a row's clusters are complete when their launch counts add up to the row's
            m = re.match(r"^(.*?)\s+n=(\d+)\s+avg=\s*([\d.]+) min=\s*([\d.]+) max=\s*([\d.]+)(?: grid=(\d+))?\s*$", l)
            if m and "pr::" in m.group(1):
                k = short(m.group(1))
                rec = {"n": int(m.group(2)), "ms_avg": float(m.group(3)), "ms_min": float(m.group(4)), "ms_max": float(m.group(5))}
                if m.group(6):
                    key = (k, int(m.group(6)))
                else:
                    key = next((kk for kk in order if kk[0] == k and sum(c["n"] for c in clusters.get(kk, [])) < trace[kk]["n"]
                                and trace[kk]["ms_min"] - 1e-9 <= rec["ms_min"] and rec["ms_max"] <= trace[kk]["ms_max"] + 1e-9), None)
                if key is not None:
                    clusters.setdefault(key, []).append(rec)
        elif mode == "pmc":
            m = re.match(r"^(.*?)\s+grid=(\d+)\s+(\S+)\s+per launch ([\d.e+-]+) \(big=(\d+)\).*?(?:\[cluster (\d+)/(\d+) ~([\d.]+) ms\])?\s*$", l)
            if m and "pr::" in m.group(1):
                ci = (int(m.group(6)), int(m.group(7))) if m.group(6) else (1, 1)
                pmc.setdefault((short(m.group(1)), int(m.group(2))), {}).setdefault(ci, {})[m.group(3)] = float(m.group(4))
    # (kernel, grid) -> list of (label, duration record, counters)
    items = []
    for key, t in trace.items():
        pc = pmc.get(key)
        if not pc:
            continue
        tcl = sorted(clusters.get(key, []), key=lambda r: r["ms_avg"])
        # PMC passes see their own dispatch mix: take, per counter, the cluster count of that pass
        ncl = max(ci[1] for ci in pc)
        if tcl and ncl == len(tcl):
            for r, tc in enumerate(tcl):
                c = {}
                for ci, cs in pc.items():
                    if ci == (r + 1, ncl):
                        c.update(cs)
                items.append(("%s @ %d [launches of ~%.2f ms]" % (key[0], key[1], tc["ms_avg"]), dict(t, **tc), c))
        else:
            c = {}
            for ci, cs in sorted(pc.items()):      # no usable clustering: the slowest cluster of every pass against the slowest of the trace
                if ci[0] == ci[1]:
                    c.update(cs)
            tt = dict(t, **tcl[-1]) if tcl else t
            items.append(("%s @ %d" % key, dict(tt, mixed=bool(tcl)), c))
    out = {"source": path, "note": "per (kernel, grid size in threads); counters are per launch (dispatches above a tenth of the largest)", "kernels": {}}
    for label, t, c in sorted(items, key=lambda it: -it[1]["ms_avg"] * it[1]["n"]):
        if not c or t["ms_avg"] < min_ms:
            continue
        e = {"ms_per_launch": t["ms_avg"], "launches_in_trace": t["n"], "vgpr": t["vgpr"], "lds_bytes": t["lds"]}
        if t.get("mixed"):      # launches of several sizes under one (kernel, grid) that the two runs did not cluster alike: rates below mix them
            e["caveat"] = "launches of different durations under this (kernel, grid) could not be matched between the trace and the PMC passes: ratios are over a mixture"
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if cyc > 0:
            e["gpu_cycles"] = cyc
            e["sustained_ghz"] = cyc / (t["ms_avg"] * 1e-3) / 1e9
            simd_cyc = cyc * 1024.0
            if c.get("SQ_INSTS_MFMA", 0) > 0:
                e["matrix_pipe_busy"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cyc
                e["valu_per_mfma"] = c.get("SQ_INSTS_VALU", 0.0) / c["SQ_INSTS_MFMA"]
                e["insts_mfma"] = c["SQ_INSTS_MFMA"]
            if "SQ_INSTS_VALU" in c:
                e["insts_valu"] = c["SQ_INSTS_VALU"]
                e["valu_issue_utilisation"] = c["SQ_INSTS_VALU"] * 4.0 / simd_cyc
            if "SQ_ACTIVE_INST_VALU" in c:
                e["valu_busy"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_cyc
            if "SQ_ACTIVE_INST_LDS" in c:
                e["lds_inst_active"] = c["SQ_ACTIVE_INST_LDS"] * 4.0 / simd_cyc
            if "SQ_LDS_BANK_CONFLICT" in c:
                e["lds_bank_conflict_cycles"] = c["SQ_LDS_BANK_CONFLICT"]
                e["lds_bank_conflict_per_cu_cycle"] = c["SQ_LDS_BANK_CONFLICT"] / (cyc * 256.0)
            if "SQ_WAIT_ANY" in c and c.get("SQ_WAVE_CYCLES", 0) > 0:
                e["wave_cycles_waiting"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        if "FETCH_SIZE" in c:
            e["hbm_bytes"] = (2.0 * c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0.0)) * 1024.0
            e["hbm_read_bytes"] = 2.0 * c["FETCH_SIZE"] * 1024.0
            e["hbm_write_bytes"] = c.get("WRITE_SIZE", 0.0) * 1024.0
            e["hbm_GBps"] = e["hbm_bytes"] / (t["ms_avg"] * 1e-3) / 1e9
            e["frac_of_8TBps"] = e["hbm_GBps"] / 8000.0
        if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) > 0:
            e["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
            e["l2_requests"] = c["TCC_HIT_sum"] + c["TCC_MISS_sum"]
        if "caveat" in e:      # a duration from one mixture of launches against counters from another: rates per second would be meaningless
            for k_ in ("sustained_ghz", "hbm_GBps", "frac_of_8TBps"):
                e.pop(k_, None)
        out["kernels"][label] = e
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
